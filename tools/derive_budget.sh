#!/bin/bash
# The derive sweep's byte budget per store group: a developer build of the library (IVX_DERIVE_DEBUG) with one store group switched off per run
# (IVX_DERIVE_SKIP), WRITE_SIZE and FETCH_SIZE of k_derive from separate rocprofv3 --pmc passes, headline and all-surface workloads.
# usage (GPU box): tools/derive_budget.sh <tag>      (rebuilds the library IN the box's copy of the tree; nothing comes back but the table)
tag=${1:-budget}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
make -C impact_amd/csrc -j16 EXTRA=-DIVX_DERIVE_DEBUG > "$out/build.log" 2>&1 || { tail -5 "$out/build.log"; exit 1; }
STEPS="--steps 6 --warmup 2 --no-cpu-baseline --no-pile --plain"
{
for wl in headline dense; do
  if [ $wl = dense ]; then W="--workload dense"; else W=""; fi
  for skip in 0 1 2 4 8 16 32 64 128 255; do
    for c in WRITE_SIZE FETCH_SIZE; do
      rm -rf "$out/p"
      IVX_DERIVE_SKIP=$skip rocprofv3 --kernel-trace --pmc $c -d "$out/p" -o p -- python3 bench.py $STEPS $W > "$out/p.log" 2>&1
      python3 - "$(ls "$out"/p/*.db | tail -1)" $wl $skip $c <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
r = list(db.execute("select avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like '%k_derive%'", (sys.argv[4],)))[0]
d = list(db.execute("select avg(end-start) from kernels where name like '%k_derive%'"))[0][0]
mult = 2.0 if sys.argv[4] == "FETCH_SIZE" else 1.0
print(f"{sys.argv[2]} skip={int(sys.argv[3]):3d} {sys.argv[4]:10s} {mult * r[0] * 1024 / 1e6:8.2f} MB per launch ({r[1]} launches) k_derive {d / 1e3:.1f} us")
PY
    done
  done
done
} 2>&1 | tee "$out/derive_budget.log"
rm -rf "$out/p"
