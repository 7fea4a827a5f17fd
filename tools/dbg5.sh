#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
export KATGPU_TESTING=1
timeout 600 python -m pytest tests/test_gpu_partition.py -x -q 2>&1 | tail -3
run() {
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['result_accounts_for_every_kmer'], d['kernel_ms_per_step'], d['roofline']['frac'])
    elif l: print(l[:300])"
}
for i in 1 2; do
echo "== noinline"; KATGPU_APPLY_NOINLINE=1 run
echo "== inline in the first round"; run
done
