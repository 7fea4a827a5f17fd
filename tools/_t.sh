cd $GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests/test_gpu_dist.py tests/test_gpu_comm.py tests/test_gpu_bench_line.py tests/test_gpu_cli.py -m gpu -x -q --timeout=1500 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/r6_t.log
