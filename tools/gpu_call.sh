#!/bin/bash
# one gpurun call of this round's A/B work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
(timeout 1200 python -m pytest tests/test_gpu_partition.py tests/test_gpu_bench_geometry.py -k "partitioned_counter or config4" -m gpu -x -q --timeout=900 --durations=8 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/c17_tests.log 2>&1
tail -16 gpurun_out/c17_tests.log | cut -c1-400
show() {
python - "$1" "$2" <<PY
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(tag, d["ms_per_step"], d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"], d.get("kernel_ms_per_step"))
except Exception as e:
    print(tag, "bench failed", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
}
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/c17_$tag.json 2> gpurun_out/c17_$tag.err
  show $tag gpurun_out/c17_$tag.json
}
run lazy1 A=1
run nolazy1 KATGPU_NO_LAZY_ZERO=1
run lazy2 A=1
run nolazy2 KATGPU_NO_LAZY_ZERO=1
