cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python tools/bench_scan_e2e.py --reads 50000000 --settings "12:32:512,16:16:256,24:32:512" > gpurun_out/r3_scan_sweep2.log 2>&1
cat gpurun_out/r3_scan_sweep2.log | cut -c1-330
(timeout 600 python -m pytest tests/test_gpu_scan.py -q --timeout=300 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r3_pytest8.log 2>&1
tail -5 gpurun_out/r3_pytest8.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench8.json 2> gpurun_out/r3_bench8.err
python3 -c "
import json; d=json.load(open('gpurun_out/r3_bench8.json')); print(d['ms_per_step']); print(json.dumps(d['end_to_end'])[:900])"
