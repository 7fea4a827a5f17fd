# the partition arena's share of the free HBM: 0.85 (four rounds at config 4) against larger shares (three rounds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/arena_ab.txt
: > $out
for f in ${FRACS:-0.85 0.60 0.45 0.85 0.60}; do
  KATGPU_ARENA_FRACTION=$f KATGPU_TRACE=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > gpurun_out/ar.json 2> gpurun_out/ar.err || tail -3 gpurun_out/ar.err >> $out
  python - $f <<'PY' >> $out 2>&1
import json, sys
j = json.loads([l for l in open("gpurun_out/ar.json") if l.startswith("{")][-1])
print("fraction", sys.argv[1], "ms_per_step", j["ms_per_step"], "kernels", json.dumps(j.get("kernel_ms_per_step")))
PY
  grep "partition round" gpurun_out/ar.err | head -1 | cut -c1-150 >> $out
  grep "partition arena of" gpurun_out/ar.err | tail -1 >> $out
done
cat $out | cut -c1-400
