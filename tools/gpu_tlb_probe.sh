#!/bin/bash
# round 6: level 1's store pattern against three buffer layouts, then the UTCL1 counters of the cases that matter
#   gpurun -- bash tools/gpu_tlb_probe.sh   ->  gpurun_out/tlb_probe.txt
set -u
root=$PWD
out=$PWD/gpurun_out
mkdir -p "$out"
timeout 600 tools/ubench_l1_tlb.bin > "$out/tlb_probe.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for c in 0 3 6 1 4; do
  rm -rf /tmp/prof_tlb
  timeout 300 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --kernel-trace --output-format csv -d /tmp/prof_tlb -- "$root/tools/ubench_l1_tlb.bin" $c > /tmp/tlb_case.txt 2> "$out/tlb_pmc.err"
  python - $c >> "$out/tlb_probe.txt" <<'PY'
import csv, glob, sys
agg = {}
n = 0
for f in glob.glob("/tmp/prof_tlb/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Counter_Name"]] = agg.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"]); n += 1
print("case", sys.argv[1], "(4 launches)", {k: "%.4g" % v for k, v in agg.items()}, open("/tmp/tlb_case.txt").read().strip())
PY
done
cat "$out/tlb_probe.txt"
