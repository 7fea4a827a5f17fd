#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
export KATGPU_TESTING=1
run() {
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e "$@" 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], round(d['value']/1e9,1), d['result_accounts_for_every_kmer'], d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline']['launches'])
    elif l: print(l[:400])"
}
echo "== hash regions"; run
echo "== minimizer regions"; KATGPU_MZ_MIN_REGIONS=4096 KATGPU_TRACE=1 run 2>&1 | grep -v "alloc keys\|table alloc" | tail -12
