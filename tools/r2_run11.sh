#!/bin/bash
out=$PWD/gpurun_out/r2o
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace -- python3 tools/time_edit.py > "$out/trace_stdout.log" 2>&1
python tools/rocpd_stats.py "$(ls "$out"/trace/*.db | tail -1)" > "$out/kernel_stats.csv"
rm -rf "$out/trace"
tail -2 "$out/trace_stdout.log"
python - "$out/kernel_stats.csv" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    print(f"{name:28s} {int(r['Calls']):4d} avg {float(r['AverageNs'])/1000:9.1f} min {float(r['MinNs'])/1000:8.1f} max {float(r['MaxNs'])/1000:8.1f} us")
PY
