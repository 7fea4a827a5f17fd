cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
(env KATGPU_TEST_SCAN_BATCH=16384 KATGPU_TEST_SCAN_SEGMENT=4096 KATGPU_TEST_SCAN_OVERLAP=2048 KATGPU_TRACE=1 timeout 200 python tests/scan_cases.py 2>&1 | grep -v "alloc\|partition round" | tail -30) > gpurun_out/r3_dbg_scan.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_comm.py tests/test_gpu_cli.py -q --timeout=300 -x -p no:cacheprovider 2>&1 | tail -60) > gpurun_out/r3_pytest6.log 2>&1
tail -30 gpurun_out/r3_dbg_scan.log | cut -c1-300; tail -30 gpurun_out/r3_pytest6.log | cut -c1-300
