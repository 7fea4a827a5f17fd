cd $GRAFT_REPO_ROOT
KATGPU_TRACE=1 timeout 300 python tools/pgz_host_bench.py --reads 24000000 --threads 8,10,12,16,24 2>&1 | grep -v '^\[katgpu +\|alloc\|context' | cut -c1-700 > gpurun_out/r6_pgz_host.txt
( time KATGPU_TRACE=1 timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-workloads > gpurun_out/r6_bench_gz.json 2> gpurun_out/r6_bench_gz.err ) 2> gpurun_out/r6_bench_gz.time
