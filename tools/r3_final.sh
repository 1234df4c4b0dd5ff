#!/bin/bash
# End-of-round run on the GPU box: the GPU test suite, the rocprofv3 traces and PMC passes of the step-only bench (tools/profile_round.sh without
# its bench line), the summaries copied into the box's profiles/round3 so that the default bench line — run LAST — quotes their traffic live, the
# slab protocol timings. Everything lands in gpurun_out/<tag>/; copy what is to be committed into profiles/round3/ afterwards.
# usage: tools/r3_final.sh <tag>
set -u
tag=${1:-r3f}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > "$out/${tag}_pytest_gpu.log" 2>&1
grep -E "passed|failed|error" "$out/${tag}_pytest_gpu.log" | tail -2
STEPS="--steps 10 --warmup 2 --no-cpu-baseline --no-pile --plain"
NST=22
for wl in headline dense; do
  if [ $wl = dense ]; then W="--workload dense"; else W=""; fi
  rocprofv3 --kernel-trace --stats -d "$out/trace_$wl" -o trace -- python3 bench.py $STEPS $W > "$out/${tag}_bench_step_only_$wl.json" 2> "$out/trace_${wl}_stderr.log"
  python tools/rocpd_stats.py "$(ls "$out"/trace_$wl/*.db | tail -1)" > "$out/${tag}_kernel_stats_step_only_$wl.csv"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch_$wl" -o fetch -- python3 bench.py $STEPS $W > "$out/pmc_fetch_$wl.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write_$wl" -o write -- python3 bench.py $STEPS $W > "$out/pmc_write_$wl.log" 2>&1
  python tools/pmc_traffic.py "$(ls "$out"/pmc_fetch_$wl/*.db | tail -1)" "$(ls "$out"/pmc_write_$wl/*.db | tail -1)" "$out/pmc_traffic_$wl.json" $NST "bench.py $STEPS $W"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d "$out/pmc_valu_$wl" -o valu -- python3 bench.py $STEPS $W > "$out/pmc_valu_$wl.log" 2>&1
  python tools/pmc_valu.py "$(ls "$out"/pmc_valu_$wl/*.db | tail -1)" "$out/pmc_valu_$wl.json" $NST
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d "$out/pmc_salu_$wl" -o salu -- python3 bench.py $STEPS $W > "$out/pmc_salu_$wl.log" 2>&1
  python tools/pmc_valu.py "$(ls "$out"/pmc_salu_$wl/*.db | tail -1)" "$out/pmc_scalar_$wl.json" $NST
  rm -rf "$out/trace_$wl" "$out/pmc_fetch_$wl" "$out/pmc_write_$wl" "$out/pmc_valu_$wl" "$out/pmc_salu_$wl"
done
mkdir -p profiles/round3
cp "$out"/pmc_traffic_*.json "$out"/pmc_valu_*.json profiles/round3/
python bench.py 2> "$out/bench_stderr.log" | tail -1 > "$out/${tag}_bench_n1.json"
for n in 1 2 4 8; do python tools/time_slab_overhead.py 200 $n 2>/dev/null | tail -2; done > "$out/${tag}_slab_overhead.log"
python tools/time_pile.py 2>/dev/null | tail -1 > "$out/${tag}_time_pile.log"
IVX_SOLVER_SERIAL=1 python tools/time_pile.py 2>/dev/null | tail -1 >> "$out/${tag}_time_pile.log"
cat "$out/${tag}_slab_overhead.log" "$out/${tag}_time_pile.log"
python - "$out/${tag}_bench_n1.json" <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms", d["ms_per_step"], "traffic", d["roofline"]["traffic"], "dense frac", d["dense"]["roofline"]["frac"], d["dense"]["roofline"].get("counter_frac"))
print("pile", d["pile"]["ms_per_step"], "frame", d["frame"]["ms_per_frame"], d["frame"]["ms_per_frame_two_streams"], d["frame"]["pipeline"]["ms_per_frame"])
P
