#!/bin/bash
# round 6: cycle stamps of the one-pass level 2 (block edition) and of the apply at a reduced config (one partition round)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
KATGPU_P2_STAMP=1 timeout 300 python bench.py --reads 100000000 --genome 300000000 --steps 2 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > gpurun_out/r6_stamp.json 2> gpurun_out/r6_stamp.err
grep -E "stamps" gpurun_out/r6_stamp.err | tail -6 | cut -c1-400
python - <<'PY'
import json
try:
    j = json.loads([l for l in open("gpurun_out/r6_stamp.json") if l.startswith("{")][-1])
    print("ms_per_step", j["ms_per_step"], "kernels", json.dumps(j.get("kernel_ms_per_step")))
except Exception as e:
    print("no bench line:", e)
PY
