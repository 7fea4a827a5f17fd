cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export KATGPU_TESTING=1
(env KATGPU_TEST_SCAN_BATCH=1048576 KATGPU_TEST_SCAN_SEGMENT=65536 KATGPU_TEST_SCAN_OVERLAP=4096 KATGPU_TRACE=1 timeout 200 python tests/scan_cases.py 2>&1 | grep -v "alloc\|partition round" | tail -12) > gpurun_out/r3_dbg_scan2.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_scan.py tests/test_gpu_cli.py -q --timeout=300 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r3_pytest9.log 2>&1
timeout 400 python tools/bench_scan_e2e.py --reads 50000000 --settings "16:8:512,12:16:512,24:8:512,16:8:1024" > gpurun_out/r3_scan_sweep3.log 2>&1
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench9.json 2> gpurun_out/r3_bench9.err
tail -12 gpurun_out/r3_dbg_scan2.log | cut -c1-250; tail -6 gpurun_out/r3_pytest9.log; cat gpurun_out/r3_scan_sweep3.log | cut -c1-300
python3 -c "
import json; d=json.load(open('gpurun_out/r3_bench9.json')); print(d['ms_per_step']); print(json.dumps(d['end_to_end'])[:700])"
