#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun from the repo root):
#   tools/profile_bench.sh <tag>      ->  gpurun_out/<tag>_bench.json, _kernel_stats.csv, _pmc_fetch_write.json,
#                                         <tag>_{hist,gcp,comp-rr}_bench.json / _kernel_stats.csv, <tag>_sq_counters.txt
# The default workload (config 4) runs three ways: plain (the JSON line, with the CPU baseline and the end-to-end leg),
# --kernel-trace --stats (per-kernel durations), and two --pmc passes (FETCH_SIZE, WRITE_SIZE; counters are collected in their own
# runs, with --kernel-trace only).  The other workloads: plain, --kernel-trace --stats, and the same two --pmc passes (tools/profile_pmc_workload.sh: the two
# counters cannot be collected in one pass on gfx950 -- rocprofv3 aborts with "exceeds the capabilities of the hardware" and then hangs).
# Config 4 from files to files at full size is the plain line's `end_to_end` leg (bench.py; tools/e2e_config4.py runs it on its own).
set -u
tag=${1:-r05_final}
root=$PWD
out=$PWD/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
quiet="--steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-workloads"
if [ "${SKIP_MAIN:-0}" != 1 ]; then        # (SKIP_MAIN=1: only the other workloads and the SQ view, after an ONLY_MAIN=1 run)
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python "$root/bench.py" $quiet > /dev/null 2> "$out/${tag}_stats.err"
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_kernel_stats.csv" \;
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof_$(echo $c | tr 'A-Z' 'a-z' | cut -d_ -f1)
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python "$root/bench.py" $quiet > /dev/null 2> "$out/${tag}_pmc_$c.err"
done
python - "$out/${tag}_pmc_fetch_write.json" "$root" <<'PY'
import csv, glob, json, re, sys
agg = {}
for d, name in (("/tmp/prof_fetch", "FETCH_SIZE"), ("/tmp/prof_write", "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
            if row["Counter_Name"] != name:
                continue
            e = agg.setdefault(k, {"launches": 0})
            e[name + "_KB_total"] = e.get(name + "_KB_total", 0.0) + float(row["Counter_Value"])
            if name == "FETCH_SIZE":
                e["launches"] += 1
sys.path.insert(0, sys.argv[2])
import bench
agg["_stage_sources_sha256_16"] = bench.stage_sources_digest()      # the kernels these counters belong to (bench.py checks it)
agg["_stage_sources"] = bench.STAGE_SOURCES
json.dump(agg, open(sys.argv[1], "w"), indent=1)
PY
# the plain line comes after the counters: its roofline.traffic is read from the profile that was just taken (same sources: bench.py checks)
mkdir -p "$root/profiles" && cp "$out/${tag}_pmc_fetch_write.json" "$root/profiles/${tag}_pmc_fetch_write.json"
timeout 900 python "$root/bench.py" --steps 3 --warmup 1 > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
fi
[ "${ONLY_MAIN:-0}" = 1 ] && { ls -la "$out" | grep "$tag"; exit 0; }
for w in hist gcp comp-rr; do
  rm -rf /tmp/prof_w /tmp/prof_wp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -- python "$root/bench.py" --workload $w $quiet > /dev/null 2> "$out/${tag}_${w}_stats.err"
  find /tmp/prof_w -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_${w}_kernel_stats.csv" \;
  bash "$root/tools/profile_pmc_workload.sh" "$tag" $w
  cp "$out/${tag}_${w}_pmc_fetch_write.json" "$root/profiles/${tag}_${w}_pmc_fetch_write.json"
  timeout 600 python "$root/bench.py" --workload $w --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > "$out/${tag}_${w}_bench.json" 2> "$out/${tag}_${w}_bench.err"
done
# (config 4 from files to files at full size is part of the plain line above since round 5: end_to_end, with its result_check)
# SQ view of the stage kernels (one partition round of a reduced config): wave cycles, waits, issue, LDS conflicts
rm -rf /tmp/prof_sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_sq -- python "$root/bench.py" --reads 60000000 --genome 200000000 $quiet > /dev/null 2> "$out/${tag}_sq.err"
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
ls -la "$out" | grep "$tag"
