cd $GRAFT_REPO_ROOT
timeout 300 python tools/exchange_components.py > gpurun_out/r6_exchange_components2.txt 2>&1
(timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_comm.py tests/test_gpu_bench_line.py -m gpu -x -q --timeout=1400 -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/r6_t.log
