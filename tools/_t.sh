cd $GRAFT_REPO_ROOT
timeout 300 python tools/pgz_host_bench.py --reads 24000000 --threads 14 --chunks-mb 2,4,8,16 2>&1 | grep -v '^\[katgpu' > gpurun_out/r6_pgz_chunks.txt
( time timeout 1200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-workloads --e2e-gz-reads 150000000 > gpurun_out/r6_bench_gz_half.json 2> gpurun_out/r6_bench_gz_half.err ) 2> gpurun_out/r6_bench_gz_half.time
