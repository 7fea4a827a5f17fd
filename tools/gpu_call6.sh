#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_partition.py tests/test_gpu_bench_geometry.py tests/test_gpu_scale_properties.py -q -x --timeout=600 --durations=6 -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/r05_l1tests.log 2>&1
tail -14 gpurun_out/r05_l1tests.log | cut -c1-250
grep -q " passed" gpurun_out/r05_l1tests.log && ! grep -q "failed" gpurun_out/r05_l1tests.log || { echo "tests failed: no bench"; exit 0; }
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], {k: v for k, v in d["kernel_ms_per_step"].items() if k.startswith("part") or k.startswith("comp")}, d["result_accounts_for_every_kmer"])
except Exception as ex:
    print(sys.argv[1], "no line:", ex, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
}
Q="--steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads"
for i in 1 2; do
  (KATGPU_TESTING=1 KATGPU_LIB_PATH=$PWD/kat_amd/libkatgpu_prev.so timeout 600 python bench.py $Q > gpurun_out/r05_ab_prev$i.json 2> gpurun_out/r05_ab_prev$i.err); summ gpurun_out/r05_ab_prev$i.json
  (timeout 600 python bench.py $Q > gpurun_out/r05_ab_new$i.json 2> gpurun_out/r05_ab_new$i.err); summ gpurun_out/r05_ab_new$i.json
done
