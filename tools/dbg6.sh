#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
export KATGPU_TESTING=1
for cfg in "8192 400000 1" "512 100000 1" "2048 250000 2"; do
  set -- $cfg
  echo "=== mz: region $1 round $2 l1_fast $3"
  KATGPU_MZ_MIN_REGIONS=2 KATGPU_TRACE=${TRACE:-} KATGPU_L1_FAST=$3 KATGPU_PART_MIN_STARTS=0 KATGPU_TEST_REGION_SLOTS=$1 KATGPU_TEST_ROUND_ITEMS=$2 KATGPU_TEST_SPILL_MOD=0 timeout 150 python tests/partition_cases.py 2>&1 | grep -v "alloc keys\|table alloc" | tail -6
done
