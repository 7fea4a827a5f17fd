cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python tools/bench_scan_e2e.py --reads 50000000 --settings "16:8:512,16:8:512" > gpurun_out/r3_scan_sweep4.log 2>&1
cat gpurun_out/r3_scan_sweep4.log | cut -c1-260
