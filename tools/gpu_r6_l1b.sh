#!/bin/bash
# round 6: the block editions of levels 1 and 2 (the defaults: `blocks`) against what they replace (`groups`: KATGPU_L1_BLOCKS=0; `p2old`: KATGPU_P2X=0) in ONE library, config 4, same call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r6_l1b.txt
: > $out
if [ -n "${TESTS:-}" ]; then (timeout 1500 python -m pytest $TESTS -m gpu -x -q --timeout=600 -p no:cacheprovider 2>&1 | tail -5) >> $out 2>&1; fi
show() { python - "$1" "$2" <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1])
    print(tag, "ms_per_step", j["ms_per_step"], "frac", j["roofline"]["frac"], "kernels", json.dumps(j.get("kernel_ms_per_step")), "distinct", j.get("distinct_table1"), j.get("result_accounts_for_every_kmer"))
except Exception as e:
    print(tag, "no line:", e)
PY
}
for v in ${ORDER:-blocks groups blocks groups}; do
  export KATGPU_TESTING=1
  unset KATGPU_L1_BLOCKS KATGPU_P2X
  if [ $v = groups ]; then export KATGPU_L1_BLOCKS=0; fi      # level 1's group edition (and k_p2_fast reading groups)
  if [ $v = p2old ]; then export KATGPU_P2X=0; fi            # level 1's blocks into k_p2_fast's block edition instead of kg_l2_blocks.hpp's kernel
  timeout 400 python bench.py ${BENCH_ARGS:-} --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > gpurun_out/l1b_$v.json 2> gpurun_out/l1b_$v.err || tail -3 gpurun_out/l1b_$v.err >> $out
  show "$v" gpurun_out/l1b_$v.json >> $out 2>&1
done
cat $out | cut -c1-700
