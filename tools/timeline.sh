#!/bin/bash
# Developer tool: the ordered kernel list (start, duration, gap to the previous kernel's end) of the tail of a python tool's run.
# usage (GPU box): tools/timeline.sh <tag> <n last kernels> <script.py> [args]
tag=$1; n=$2; shift; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$out/trace" -o trace -- python3 "$@" > "$out/stdout.log" 2>&1
python3 - "$(ls "$out"/trace/*.db | tail -1)" "$n" > "$out/timeline.txt" <<'PY'
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))[-int(sys.argv[2]):]
prev = None
t00 = rows[0][1]
for name, s, e in rows:
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{(s - t00) / 1e3:10.1f} us  {name:30s} {(e - s) / 1e3:8.1f} us   gap {gap:8.1f}")
    prev = e
PY
rm -rf "$out/trace"
tail -3 "$out/stdout.log"
cat "$out/timeline.txt"
