cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_gpu_comm.py tests/test_gpu_bench_line.py -m gpu -q --timeout=200 -p no:cacheprovider --durations=5 2>&1 | tail -25) > gpurun_out/one_test.log 2>&1
tail -14 gpurun_out/one_test.log | cut -c1-200
