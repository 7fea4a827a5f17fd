cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -40) > gpurun_out/r3_pytest1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_pytest1.log
timeout 400 python bench.py --steps 5 --warmup 1 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err
for v in "KATGPU_JOIN_BLOCK=1024" "KATGPU_APPLY_PER_CU=1" "KATGPU_NO_FOLD=1" "KATGPU_NO_PACKED=1"; do
  env KATGPU_TESTING=1 $v timeout 300 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r3_ab_$v.json 2> gpurun_out/r3_ab_$v.err
done
tail -3 gpurun_out/r3_pytest1.log; cat gpurun_out/r3_bench1.json | cut -c1-1500
