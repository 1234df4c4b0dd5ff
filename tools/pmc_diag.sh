#!/bin/bash
# diagnostic counters per kernel (averages per launch), one rocprofv3 --pmc pass per group of counters
# usage (GPU box): tools/pmc_diag.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d "$out/p$i" -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pile --plain ${IVX_DIAG_ARGS:-} > "$out/p$i.log" 2>&1
  python - "$(ls "$out"/p$i/*.db | tail -1)" <<'PY'
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name")
tab = collections.defaultdict(dict)
for k, c, v in rows:
    k = re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "").replace("void ", ""))
    tab[k][c] = v
names = sorted({c for d in tab.values() for c in d})
print("kernel".ljust(24), *[n[:18].rjust(18) for n in names])
for k in sorted(tab):
    print(k[:24].ljust(24), *[f"{tab[k].get(n, 0):18.1f}" for n in names])
PY
  rm -rf "$out/p$i"
done
