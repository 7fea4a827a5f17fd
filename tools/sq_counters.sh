#!/bin/bash
# SQ view of the stage kernels (one partition round of a reduced config): wave cycles, waits, issue, LDS conflicts.
#   tools/sq_counters.sh <tag>   ->  gpurun_out/<tag>_sq_counters.txt      (run through gpurun from the repo root)
set -u
tag=${1:-sq}
root=$PWD
out=$PWD/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_sq -- python "$root/bench.py" --reads 60000000 --genome 200000000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> "$out/${tag}_sq.err"
python - "$out/${tag}_sq_counters.txt" <<'PY'
import csv, glob, re, sys
agg = {}
for f in glob.glob("/tmp/prof_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"[<(].*", "", row["Kernel_Name"]).replace("void ", "").replace("kg::", "").strip()
        e = agg.setdefault(k, {})
        e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
with open(sys.argv[1], "w") as o:
    o.write("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace\n")
    o.write("bench.py --reads 60000000 --genome 200000000 --steps 1 --warmup 0 (one partition round); fractions of SQ_WAVE_CYCLES\n")
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        w = e.get("SQ_WAVE_CYCLES", 0) or 1
        o.write("%-28s wave_cycles=%.3g wait_any=%.2f wait_inst_any=%.2f active_inst_any=%.2f active_valu=%.2f active_lds=%.2f lds_bank_conflict=%.3g lds_idx_active=%.3g\n" % (
            k, w, e.get("SQ_WAIT_ANY", 0) / w, e.get("SQ_WAIT_INST_ANY", 0) / w, e.get("SQ_ACTIVE_INST_ANY", 0) / w, e.get("SQ_ACTIVE_INST_VALU", 0) / w,
            e.get("SQ_ACTIVE_INST_LDS", 0) / w, e.get("SQ_LDS_BANK_CONFLICT", 0), e.get("SQ_LDS_IDX_ACTIVE", 0)))
PY
cat "$out/${tag}_sq_counters.txt"
