#!/bin/bash
# round 5, first GPU call: the new tests, then the driver's bench command at a short step count (the whole new line: workloads, full-size e2e, CPU sample)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_comm.py tests/test_gpu_bench_line.py tests/test_gpu_ingest_at_size.py -q -x --timeout=900 --durations=25 -p no:cacheprovider -s 2>&1 | tail -120) > gpurun_out/r05_newtests.log 2>&1
tail -40 gpurun_out/r05_newtests.log | cut -c1-300
(timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err); echo "bench rc=$?"
tail -c 1500 gpurun_out/r05_bench_a.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_bench_a.json").read().strip().splitlines()[-1])
    e = d["end_to_end"]
    print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernels", d["kernel_ms_per_step"])
    print("e2e", {k: e.get(k) for k in ("value", "seconds", "full_size", "result_check", "result_check_detail", "files_written_in_s", "error")})
    print("workloads", {k: (v.get("ms_per_step"), v.get("roofline", {}).get("frac"), v.get("result_accounts_for_every_kmer"), v.get("error")) for k, v in d.get("workloads", {}).items()})
    print("cpu", d["cpu_baseline"])
except Exception as ex:
    print("no line:", ex)
PY
