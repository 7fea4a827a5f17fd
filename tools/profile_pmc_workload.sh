#!/bin/bash
# FETCH_SIZE and WRITE_SIZE per kernel for one bench.py workload, in two separate --pmc passes (one counter each, --kernel-trace only):
#   tools/profile_pmc_workload.sh <tag> <workload>   ->  gpurun_out/<tag>_<workload>_pmc_fetch_write.json
set -u
tag=$1; w=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_wp_$c
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_wp_$c -- python "$root/bench.py" --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2> "$out/${tag}_${w}_pmc_$c.err"
done
python - "$out/${tag}_${w}_pmc_fetch_write.json" "$root" <<'PY'
import csv, glob, json, re, sys
agg = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("/tmp/prof_wp_%s/**/*counter_collection.csv" % name, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != name:
                continue
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
            e = agg.setdefault(k, {"launches": 0})
            e[name + "_KB_total"] = e.get(name + "_KB_total", 0.0) + float(row["Counter_Value"])
            if name == "FETCH_SIZE":
                e["launches"] += 1
sys.path.insert(0, sys.argv[2])
import bench
agg["_stage_sources_sha256_16"] = bench.stage_sources_digest()      # the kernels these counters belong to (bench.py checks it)
json.dump(agg, open(sys.argv[1], "w"), indent=1)
PY
