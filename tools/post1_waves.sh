#!/bin/bash
# developer experiment: k_step_post1 compiled for 8 / 7 / 6 waves per SIMD (64 / 72 / 80 registers), its launch time on the headline and
# all-surface workloads and on the edit path. usage (GPU box): tools/post1_waves.sh <tag>
tag=${1:-p1w}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
STEPS="--steps 10 --warmup 3 --no-cpu-baseline --no-pile --plain"
{
for w in 8 7 6; do
  make -C impact_amd/csrc -j16 EXTRA=-DIVX_POST1_WAVES=$w > "$out/build_$w.log" 2>&1 || { tail -5 "$out/build_$w.log"; exit 1; }
  for wl in headline dense; do
    if [ $wl = dense ]; then W="--workload dense"; else W=""; fi
    rm -rf "$out/p"
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/p" -o p -- python3 bench.py $STEPS $W > "$out/p.json" 2> "$out/p.err"
    python3 - "$(ls "$out"/p/*.db | tail -1)" $wl $w <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
r = list(db.execute("select avg(value) from counters_collection where counter_name = 'WRITE_SIZE' and kernel_name like '%k_step_post1%'"))[0][0]
d = list(db.execute("select avg(end-start), min(end-start) from kernels where name like '%k_step_post1%'"))[0]
print(f"waves {sys.argv[3]} {sys.argv[2]:9s} k_step_post1 avg {d[0] / 1e3:6.1f} us min {d[1] / 1e3:6.1f} us, writes {r * 1024 / 1e6:7.2f} MB per launch")
PY
    python3 bench.py $STEPS $W 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('      unprofiled: ms/step', round(d['ms_per_step'],4), 'post1', d['stage_ms']['post1'])"
  done
  python3 tools/time_edit.py 2>&1 | grep -i "C call\|edit" | tail -3
done
} 2>&1 | tee "$out/post1_waves.log"
rm -rf "$out/p"
