#!/bin/bash
# round 6: same-call A/B of the library as built (new) against kat_amd/libkatgpu_prev.so (round 5's sources), config 4, 3 steps each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r6_ab.txt
: > $out
if [ -n "${TESTS:-}" ]; then (timeout 900 python -m pytest $TESTS -m gpu -x -q --timeout=300 -p no:cacheprovider 2>&1 | tail -5) >> $out 2>&1; fi
show() { python - "$1" "$2" <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1])
    print(tag, "ms_per_step", j["ms_per_step"], "frac", j["roofline"]["frac"], "kernels", json.dumps(j.get("kernel_ms_per_step")))
except Exception as e:
    print(tag, "no line:", e)
PY
}
for lib in ${ORDER:-new prev new prev}; do
  if [ $lib = prev ]; then export KATGPU_TESTING=1 KATGPU_LIB_PATH=$PWD/kat_amd/libkatgpu_prev.so; else unset KATGPU_TESTING KATGPU_LIB_PATH; fi
  timeout 400 python bench.py ${BENCH_ARGS:-} --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > gpurun_out/ab_$lib.json 2> gpurun_out/ab_$lib.err || tail -3 gpurun_out/ab_$lib.err >> $out
  show "$lib" gpurun_out/ab_$lib.json >> $out 2>&1
done
unset KATGPU_LIB_PATH
if [ -n "${STAMP:-}" ]; then
  KATGPU_TESTING=1 KATGPU_P2_STAMP=1 timeout 300 python bench.py --reads 100000000 --genome 300000000 --steps 1 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > /dev/null 2> gpurun_out/r6_stamp.err
  grep -E "stamps" gpurun_out/r6_stamp.err | tail -3 | cut -c1-300 >> $out
fi
cat $out | cut -c1-600
