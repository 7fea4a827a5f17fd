cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 500 python tools/bench_scan_e2e.py --reads 50000000 > gpurun_out/r3_scan_sweep.log 2>&1
cat gpurun_out/r3_scan_sweep.log | cut -c1-400
