#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun from the repo root):
#   tools/profile_bench.sh <tag>      ->  gpurun_out/<tag>_bench.json, _kernel_stats.csv, _pmc_fetch_write.json
# Three runs of the same command: plain (the JSON line), --kernel-trace --stats (per-kernel durations), and two --pmc passes
# (FETCH_SIZE, WRITE_SIZE; counters are collected in their own runs, with --kernel-trace only).
set -u
tag=${1:-r01_final}
out=$PWD/gpurun_out
mkdir -p "$out"
cmd="python $PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
timeout 600 python "$OLDPWD/bench.py" --steps 3 --warmup 1 > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$out/${tag}_stats.err"
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_kernel_stats.csv" \;
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof_$(echo $c | tr 'A-Z' 'a-z' | cut -d_ -f1)
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$out/${tag}_pmc_$c.err"
done
python - "$out/${tag}_pmc_fetch_write.json" <<'PY'
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
json.dump(agg, open(sys.argv[1], "w"), indent=1)
PY
ls -la "$out" | grep "$tag"
