# A/B of the reducers: the library as built (new) against kat_amd/libkatgpu_prev.so (the previous build), same box, same seeds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/comp_ab.txt
: > $out
(timeout 900 python -m pytest tests -m gpu -q --timeout=300 -k "${KSEL:-comp}" -p no:cacheprovider 2>&1 | tail -8) >> $out 2>&1
show() { python - "$1" "$2" <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
line = [l for l in open(path) if l.startswith("{")][-1]
j = json.loads(line)
print(tag, "ms_per_step", j["ms_per_step"], "reducers", json.dumps(j.get("reducers")), "kernels", json.dumps(j.get("kernel_ms_per_step")))
PY
}
for wl in ${WLS:-"" "--workload comp-rr"}; do
  for lib in new prev new prev; do
    if [ $lib = prev ]; then export KATGPU_TESTING=1 KATGPU_LIB_PATH=$PWD/kat_amd/libkatgpu_prev.so; else unset KATGPU_TESTING KATGPU_LIB_PATH; fi
    timeout 400 python bench.py $wl --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > gpurun_out/ab_$lib.json 2> gpurun_out/ab_$lib.err || tail -3 gpurun_out/ab_$lib.err >> $out
    show "$lib $wl" gpurun_out/ab_$lib.json >> $out 2>&1
  done
done
cat $out | cut -c1-900
