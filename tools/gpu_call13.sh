#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
Q="--steps 2 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads"
for sg in 0 127 254 60; do
  (KATGPU_L1_STAGGER=$sg timeout 600 python bench.py $Q > gpurun_out/r05_sg_$sg.json 2> gpurun_out/r05_sg_$sg.err)
  python - gpurun_out/r05_sg_$sg.json $sg <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("stagger", sys.argv[2], "ms_per_step", d["ms_per_step"], {k: v for k, v in d["kernel_ms_per_step"].items() if k.startswith("part")}, d["result_accounts_for_every_kmer"])
except Exception as ex:
    print("stagger", sys.argv[2], "no line:", ex)
PY
done
