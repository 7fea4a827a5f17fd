cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1000 python -m pytest tests -m gpu -q --timeout=300 --durations=15 -p no:cacheprovider 2>&1 | tail -80) > gpurun_out/r3_pytest4.log 2>&1
tail -30 gpurun_out/r3_pytest4.log
