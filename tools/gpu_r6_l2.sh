#!/bin/bash
# round 6: the block edition of level 2 -- parity of the partition paths, then the bench step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_partition.py tests/test_gpu_bench_geometry.py -m gpu -x -q --timeout=300 -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/r6_l2_tests.log 2>&1
tail -8 gpurun_out/r6_l2_tests.log | cut -c1-300
timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > gpurun_out/r6_l2_bench.json 2> gpurun_out/r6_l2_bench.err
tail -3 gpurun_out/r6_l2_bench.err | cut -c1-300
python - <<'PY'
import json
try:
    j = json.loads([l for l in open("gpurun_out/r6_l2_bench.json") if l.startswith("{")][-1])
    print("ms_per_step", j["ms_per_step"], "kernels", json.dumps(j.get("kernel_ms_per_step")), "frac", j["roofline"]["frac"])
except Exception as e:
    print("no bench line:", e)
PY
