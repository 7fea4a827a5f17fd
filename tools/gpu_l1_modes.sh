# level 1 measured 137 or 176 ms per step in different processes of one box: which runs, and what the clocks say meanwhile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/l1_modes.txt
: > $out
(while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n' | cut -c1-600; echo; sleep 1; done) > gpurun_out/smi.log 2>&1 &
smi=$!
for run in plain plain trace plain trace; do
  if [ $run = trace ]; then export KATGPU_TRACE=1; else unset KATGPU_TRACE; fi
  date +%s.%N >> $out
  timeout 400 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > gpurun_out/m.json 2> gpurun_out/m.err || tail -3 gpurun_out/m.err >> $out
  python - $run <<'PY' >> $out 2>&1
import json, sys
j = json.loads([l for l in open("gpurun_out/m.json") if l.startswith("{")][-1])
print(sys.argv[1], "ms_per_step", j["ms_per_step"], "kernels", json.dumps(j.get("kernel_ms_per_step")))
PY
done
kill $smi
cat $out
python - <<'PY'
import json, re
rows = []
for line in open("gpurun_out/smi.log"):
    try: j = json.loads(line)
    except Exception: continue
    c = j.get("card0", {})
    rows.append((c.get("sclk clock speed:"), c.get("mclk clock speed:"), c.get("Current Socket Graphics Package Power (W)") or c.get("Average Graphics Package Power (W)")))
print(len(rows), "smi samples; distinct:", sorted(set(rows))[:40])
PY
head -c 700 gpurun_out/smi.log
