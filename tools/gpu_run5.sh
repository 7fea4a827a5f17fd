cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_comm.py tests/test_gpu_cli.py -q --timeout=300 --durations=8 -p no:cacheprovider 2>&1 | tail -120) > gpurun_out/r3_pytest5.log 2>&1
free -g | head -2 > gpurun_out/r3_box.txt; df -h /tmp | tail -1 >> gpurun_out/r3_box.txt; nproc >> gpurun_out/r3_box.txt
timeout 800 python bench.py --steps 3 --warmup 1 > gpurun_out/r3_bench5.json 2> gpurun_out/r3_bench5.err
tail -25 gpurun_out/r3_pytest5.log; cat gpurun_out/r3_box.txt; tail -3 gpurun_out/r3_bench5.err
