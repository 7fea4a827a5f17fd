cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1 KATGPU_PART_MIN_STARTS=0
(env KATGPU_TEST_REGION_SLOTS=1024 KATGPU_TEST_ROUND_ITEMS=3000000 KATGPU_TEST_SPILL_MOD=7 KATGPU_TRACE=1 timeout 200 python tests/partition_cases.py 2>&1 | tail -40) > gpurun_out/r3_dbg62.log 2>&1
(env KATGPU_TEST_REGION_SLOTS=1024 KATGPU_TEST_ROUND_ITEMS=3000000 KATGPU_TEST_SPILL_MOD=0 timeout 200 python tests/partition_cases.py 2>&1 | tail -8) > gpurun_out/r3_dbg62b.log 2>&1
tail -12 gpurun_out/r3_dbg62.log; tail -5 gpurun_out/r3_dbg62b.log
