cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=6 --timeout=600 2>&1 | tail -60) > gpurun_out/r3_pytest2.log 2>&1
for v in "KATGPU_X=0" "KATGPU_APPLY_BLOCK=1024" "KATGPU_NO_FUSED=1"; do
  env KATGPU_TESTING=1 $v timeout 300 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r3_ab2_$v.json 2> gpurun_out/r3_ab2_$v.err
done
tail -5 gpurun_out/r3_pytest2.log
