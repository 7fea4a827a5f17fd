#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for cfg in "8192 400000" "512 100000" "2048 250000"; do
  set -- $cfg
  echo "=== APPLY_V=2 region $1 round $2"
  KATGPU_APPLY_V=2 KATGPU_PART_MIN_STARTS=0 KATGPU_TEST_REGION_SLOTS=$1 KATGPU_TEST_ROUND_ITEMS=$2 KATGPU_TEST_SPILL_MOD=0 timeout 120 python tests/partition_cases.py 2>&1 | tail -3
done
bash tools/ab_apply.sh "2 43;2 1043;2 83;2 84"
