#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_feeder.py tests/test_gpu_dist.py tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_cli.py tests/test_gpu_scale_properties.py tests/test_gpu_partition.py -x -q 2>&1 | tail -8
for jb in 512 1024; do
echo "== workload comp (default), join block $jb"
KATGPU_JOIN_BLOCK=$jb timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['result_accounts_for_every_kmer'], d['kernel_ms_per_step'], d['reducers'])
    elif l: print(l[:300])"
done
