cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
df -h /dev/shm | tail -1
timeout 900 python tools/e2e_config4.py > gpurun_out/r3_e2e_config4.json 2> gpurun_out/r3_e2e_config4.err
cut -c1-1800 gpurun_out/r3_e2e_config4.json; tail -3 gpurun_out/r3_e2e_config4.err
# two ranks sharing this GPU through bench.py's native path (SHM transport), reduced size
env KATGPU_COMM_TRANSPORT=shm HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --reads 20000000 --genome 100000000 > gpurun_out/r3_bench_g2.json 2> gpurun_out/r3_bench_g2.err
cut -c1-1500 gpurun_out/r3_bench_g2.json; tail -5 gpurun_out/r3_bench_g2.err | cut -c1-300
