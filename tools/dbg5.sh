#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_partition.py tests/test_gpu_feeder.py -x -q 2>&1 | tail -6
for f in 0 1; do
echo "== KATGPU_L1_FAST=$f"
KATGPU_L1_FAST=$f timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['result_accounts_for_every_kmer'], d['kernel_ms_per_step'], d['roofline']['frac'])
    elif l: print(l[:300])"
done
