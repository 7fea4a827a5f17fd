cd $GRAFT_REPO_ROOT
(timeout 3300 python -m pytest tests -m gpu -q --timeout=1500 -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/r6_gpu_tests.log
