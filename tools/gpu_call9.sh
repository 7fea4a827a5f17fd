#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_partition.py -q -x --timeout=600 --durations=4 -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/r05_blk_tests.log 2>&1
tail -12 gpurun_out/r05_blk_tests.log | cut -c1-300
grep -q " passed" gpurun_out/r05_blk_tests.log && ! grep -q "failed" gpurun_out/r05_blk_tests.log || { echo "tests failed: no bench"; exit 0; }
(timeout 600 python -m pytest tests/test_gpu_bench_geometry.py -q -x --timeout=600 -p no:cacheprovider -k "prefix or partitioned" 2>&1 | tail -5) > gpurun_out/r05_blk_tests2.log 2>&1
tail -5 gpurun_out/r05_blk_tests2.log | cut -c1-300
grep -q " passed" gpurun_out/r05_blk_tests2.log && ! grep -q "failed" gpurun_out/r05_blk_tests2.log || { echo "geometry tests failed: no bench"; exit 0; }
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], {k: v for k, v in d["kernel_ms_per_step"].items() if k.startswith("part") or k.startswith("comp") or k == "count"}, d["result_accounts_for_every_kmer"])
except Exception as ex:
    print(sys.argv[1], "no line:", ex, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
}
Q="--steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads"
export KATGPU_TESTING=1
for i in 1 2; do
  (KATGPU_L1_BLOCKS=0 timeout 600 python bench.py $Q > gpurun_out/r05_blk_off$i.json 2> gpurun_out/r05_blk_off$i.err); summ gpurun_out/r05_blk_off$i.json
  (timeout 600 python bench.py $Q > gpurun_out/r05_blk_on$i.json 2> gpurun_out/r05_blk_on$i.err); summ gpurun_out/r05_blk_on$i.json
done
