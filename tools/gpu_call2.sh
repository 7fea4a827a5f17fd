#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 tools/ubench_l2_layout.bin) > gpurun_out/r05_ubench_l2.txt 2>&1; cat gpurun_out/r05_ubench_l2.txt
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], {k: v for k, v in d["kernel_ms_per_step"].items() if k.startswith("part") or k.startswith("comp")})
    e = d.get("end_to_end")
    if e:
        print("  e2e", {k: e.get(k) for k in ("value", "seconds", "full_size", "result_check", "error")})
        b = e.get("breakdown", {})
        print("  ", b.get("phases"), "unaccounted", b.get("unaccounted_ms"))
        for f in b.get("files", []): print("   ", {k: f[k] for k in ("file", "setup_ms", "wall_ms", "reader_wait_ms", "scan_ms", "counter_wait_ms")})
        for l in b.get("alloc_trace", []): print("   ", l)
except Exception as ex:
    print(sys.argv[1], "no line:", ex)
PY
}
(timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > gpurun_out/r05_b_noe2e.json 2> gpurun_out/r05_b_noe2e.err); summ gpurun_out/r05_b_noe2e.json
(timeout 900 python -m pytest tests/test_gpu_bench_line.py tests/test_gpu_ingest_at_size.py -q -x --timeout=900 --durations=10 -p no:cacheprovider -s 2>&1 | tail -60) > gpurun_out/r05_newtests2.log 2>&1
tail -25 gpurun_out/r05_newtests2.log | cut -c1-300
(timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-workloads > gpurun_out/r05_b_e2e.json 2> gpurun_out/r05_b_e2e.err); summ gpurun_out/r05_b_e2e.json
(timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-workloads > gpurun_out/r05_b_noe2e2.json 2> gpurun_out/r05_b_noe2e2.err); summ gpurun_out/r05_b_noe2e2.json
