cd $GRAFT_REPO_ROOT
timeout 300 python tools/exchange_components.py > gpurun_out/r6_exchange_components.txt 2>&1
( time timeout 1200 python bench.py --steps 3 --warmup 1 > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err ) 2> gpurun_out/r06_final_bench.time
