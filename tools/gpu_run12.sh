cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_scan.py tests/test_gpu_cli.py tests/test_gpu_feeder.py -q --timeout=300 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r3_pytest12.log 2>&1
tail -5 gpurun_out/r3_pytest12.log
timeout 900 python tools/e2e_config4.py > gpurun_out/r3_e2e_config4.json 2> gpurun_out/r3_e2e_config4.err
python3 -c "
import json; d=json.load(open('gpurun_out/r3_e2e_config4.json')); print(d['seconds'], d['input_GB_per_s'], d['kmers_per_s']); print('\n'.join(d['trace'])[:3000]); print(d['phases'])"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench12.json 2> gpurun_out/r3_bench12.err
python3 -c "
import json; d=json.load(open('gpurun_out/r3_bench12.json')); print(d['ms_per_step']); print(json.dumps(d['end_to_end'])[:600])"
