#!/bin/bash
# one gpurun call of this round's work (scratch; rewritten per call)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q --timeout=900 --durations=6 -p no:cacheprovider --deselect tests/test_gpu_partition.py --deselect tests/test_gpu_bench_geometry.py 2>&1 | tail -30) > gpurun_out/c18_tests.log 2>&1
tail -14 gpurun_out/c18_tests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
