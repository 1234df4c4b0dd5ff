#!/bin/bash
# per-kernel durations of an arbitrary python tool: tools/kernel_trace_cmd.sh <tag> <script.py> [args]
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace -- python3 "$@" > "$out/trace_stdout.log" 2>&1
python tools/rocpd_stats.py "$(ls "$out"/trace/*.db | tail -1)" > "$out/kernel_stats.csv"
rm -rf "$out/trace"
python - "$out/kernel_stats.csv" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    print(f"{name:28s} {int(r['Calls']):4d} {float(r['AverageNs'])/1000:9.1f} us")
PY
