#!/bin/bash
# one gpurun call of this round's A/B work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
# reader micro-benchmark: 8 GB of file in /dev/shm
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/reader_bench.hip -o /tmp/reader_bench -lpthread 2>/dev/null
  python - <<'PY'
import numpy as np
b = np.random.default_rng(1).integers(65, 85, 64 << 20, dtype=np.uint8).tobytes()
with open('/dev/shm/katgpu_reader_bench.bin', 'wb') as f:
    for _ in range(128): f.write(b)
PY
  timeout 200 /tmp/reader_bench /dev/shm/katgpu_reader_bench.bin 16 48; rm -f /dev/shm/katgpu_reader_bench.bin; lscpu | grep -E "Model name|Socket|NUMA node|^CPU\(s\)"; numactl -H 2>/dev/null | head -8 ) > gpurun_out/c4_reader.txt 2>&1
cat gpurun_out/c4_reader.txt
(timeout 900 python -m pytest tests/test_gpu_partition.py tests/test_gpu_bench_geometry.py -m gpu -x -q --timeout=400 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/c4_tests.log 2>&1
tail -6 gpurun_out/c4_tests.log | cut -c1-300
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c4_bench.json").read().strip().splitlines()[-1])
    print("config4", d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"], d["roofline"]["launches"])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/c4_bench.err").read()[-1500:])
PY
timeout 300 python bench.py --workload comp-rr --steps 2 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/c4_bench_rr.json 2> gpurun_out/c4_bench_rr.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c4_bench_rr.json").read().strip().splitlines()[-1])
    print("comp-rr", d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"])
except Exception as e:
    print("rr bench failed", e); print(open("gpurun_out/c4_bench_rr.err").read()[-1500:])
PY
