cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "KATGPU_APPLY_NR=2" "KATGPU_APPLY_INLINE=1" "KATGPU_P1_WGS=2"; do
  env KATGPU_TESTING=1 $v timeout 300 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r3_ab3_$v.json 2> gpurun_out/r3_ab3_$v.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r3_ab3_$v.json')); print('$v', d['ms_per_step'], d['kernel_ms_per_step'])"
done
