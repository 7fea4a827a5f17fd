cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -q --timeout=300 --durations=10 -p no:cacheprovider 2>&1 | tail -60) > gpurun_out/r3_suite.log 2>&1
tail -25 gpurun_out/r3_suite.log | cut -c1-200
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) >> gpurun_out/r3_suite.log
tail -2 gpurun_out/r3_suite.log
