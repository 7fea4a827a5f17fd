#!/bin/bash
# one gpurun call of this round's A/B work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
(timeout 600 python -m pytest tests/test_gpu_partition.py -m gpu -x -q --timeout=600 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/c12_tests.log 2>&1
tail -4 gpurun_out/c12_tests.log | cut -c1-300
show() {
python - "$1" "$2" <<PY
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    pk = d["roofline"].get("per_kernel", {})
    print(tag, d["ms_per_step"], d.get("result_accounts_for_every_kmer"), d["roofline"]["frac"], d.get("kernel_ms_per_step"), {k: v.get("launches") for k, v in pk.items()})
except Exception as e:
    print(tag, "bench failed", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
}
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/c12_$tag.json 2> gpurun_out/c12_$tag.err
  show $tag gpurun_out/c12_$tag.json
  grep "level-2 stamps" gpurun_out/c12_$tag.err | head -2
}
cp kat_amd/libkatgpu.so /tmp/def.so
use() { cp $1 kat_amd/libkatgpu.so; }
run def1 A=1
use kat_amd/libkatgpu_n12b.so
run n12b_pf0 A=1
run n12b_pf2 KATGPU_P2_PF=2
run n12b_pf3 KATGPU_P2_PF=3
use kat_amd/libkatgpu_n8s.so
run n8s_pf0 A=1
run n8s_pf2 KATGPU_P2_PF=2
run n8s_pf3 KATGPU_P2_PF=3
use kat_amd/libkatgpu_n12s.so
run n12s_pf0 A=1
use /tmp/def.so
run def2 A=1
use kat_amd/libkatgpu_n12b.so
run n12b_pf2_stamp KATGPU_P2_PF=2 KATGPU_P2_STAMP=1
use kat_amd/libkatgpu_n8s.so
run n8s_pf2_stamp KATGPU_P2_PF=2 KATGPU_P2_STAMP=1
use /tmp/def.so
