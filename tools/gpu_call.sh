#!/bin/bash
# one gpurun call of this round's A/B work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
(KATGPU_COMP_PLAIN_INC=1 timeout 600 python -m pytest tests/test_gpu_comp_forms.py tests/test_gpu_parity.py -m gpu -x -q --timeout=600 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/c20_tests.log 2>&1
tail -4 gpurun_out/c20_tests.log | cut -c1-400
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
run() {  # tag, workload, env...
  tag=$1; w=$2; shift; shift
  env "$@" timeout 400 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/c20_$tag.json 2> gpurun_out/c20_$tag.err
  show $tag gpurun_out/c20_$tag.json
}
run agg1 comp A=1
run plain1 comp KATGPU_COMP_PLAIN_INC=1
run agg2 comp A=1
run plain2 comp KATGPU_COMP_PLAIN_INC=1
run gcp_agg gcp A=1
run gcp_plain gcp KATGPU_COMP_PLAIN_INC=1
run rr_agg comp-rr A=1
run rr_plain comp-rr KATGPU_COMP_PLAIN_INC=1
