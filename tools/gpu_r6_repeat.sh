#!/bin/bash
# round 6: the same command in N consecutive processes on one box (the verdict's "10 processes within 2 %"): config 4, 3 steps each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r6_repeat.txt
: > $out
for i in $(seq 1 ${N:-10}); do
  timeout 400 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > gpurun_out/rep.json 2> gpurun_out/rep.err || tail -3 gpurun_out/rep.err >> $out
  python - $i <<'PY' >> $out
import json, sys
j = json.loads([l for l in open("gpurun_out/rep.json") if l.startswith("{")][-1])
print("process", sys.argv[1], "ms_per_step", j["ms_per_step"], "frac", j["roofline"]["frac"], json.dumps(j["roofline"]["ms_per_step"]))
PY
done
python - <<'PY' >> $out
import re
rows = [l for l in open("gpurun_out/r6_repeat.txt") if l.startswith("process")]
for key in ("ms_per_step", '"l1"', '"l2"', '"apply"'):
    v = [float(re.search(re.escape(key) + r'[": ]+([0-9.]+)', l).group(1)) for l in rows]
    print("%-12s min %.1f max %.1f spread %.1f %%" % (key, min(v), max(v), 100 * (max(v) - min(v)) / min(v)))
PY
cat $out
