#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
Q="--steps 1 --warmup 0 --no-e2e --no-cpu-baseline --no-workloads --reads 100000000"
for xm in 0 1 2; do
  (KATGPU_L1_XMODE=$xm KATGPU_L1_STAMP=1 timeout 300 python bench.py $Q > gpurun_out/r05_xm_$xm.json 2> gpurun_out/r05_xm_$xm.err)
  echo "xmode $xm"; grep -a "stamps" gpurun_out/r05_xm_$xm.err | head -1 | cut -c1-330
done
