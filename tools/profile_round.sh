#!/bin/bash
# Runs on the GPU box (via gpurun): the default bench line, the rocprofv3 kernel trace of the step-only bench command (headline and
# all-surface workloads) and the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ VALU counters; separate runs, no other trace domains).
# Summaries land in gpurun_out/<tag>/ and are copied into profiles/roundN/ by hand afterwards (they carry the hash of the kernel sources: tools/source_sha.py).
# usage: tools/profile_round.sh <tag> [skip-tests]
set -u
tag=${1:-run}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
if [ "${2:-}" != "skip-tests" ]; then
  python -m pytest tests -m gpu -x -q > "$out/${tag}_pytest_gpu.log" 2>&1
  tail -2 "$out/${tag}_pytest_gpu.log"
fi
# --plain: full steps only: 2 warm-up + 10 timed + 10 with every slot timed = 22 steps
STEPS="--steps 10 --warmup 2 --no-cpu-baseline --no-pile --plain"
NST=22
for wl in headline dense; do
  if [ $wl = dense ]; then W="--workload dense"; else W=""; fi
  rocprofv3 --kernel-trace --stats -d "$out/trace_$wl" -o trace -- python3 bench.py $STEPS $W > "$out/${tag}_bench_step_only_$wl.json" 2> "$out/trace_${wl}_stderr.log"
  python tools/rocpd_stats.py "$(ls "$out"/trace_$wl/*.db | tail -1)" > "$out/${tag}_kernel_stats_step_only_$wl.csv"
  python tools/kernel_share.py "$out/${tag}_kernel_stats_step_only_$wl.csv" "$out/kernel_share_$wl.json" > /dev/null
  # ... and the step as a step of its own (no pre-pass a step ahead beside the other kernels): the kernels' durations undisturbed
  IVX_BENCH_SAMPLE_AHEAD=0 rocprofv3 --kernel-trace --stats -d "$out/trace_iso_$wl" -o trace -- python3 bench.py $STEPS $W > "$out/${tag}_bench_step_only_isolated_$wl.json" 2> "$out/trace_iso_${wl}_stderr.log"
  python tools/rocpd_stats.py "$(ls "$out"/trace_iso_$wl/*.db | tail -1)" > "$out/${tag}_kernel_stats_step_only_isolated_$wl.csv"
  rm -rf "$out/trace_iso_$wl"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch_$wl" -o fetch -- python3 bench.py $STEPS $W > "$out/pmc_fetch_$wl.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write_$wl" -o write -- python3 bench.py $STEPS $W > "$out/pmc_write_$wl.log" 2>&1
  python tools/pmc_traffic.py "$(ls "$out"/pmc_fetch_$wl/*.db | tail -1)" "$(ls "$out"/pmc_write_$wl/*.db | tail -1)" "$out/pmc_traffic_$wl.json" $NST "bench.py $STEPS $W"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d "$out/pmc_valu_$wl" -o valu -- python3 bench.py $STEPS $W > "$out/pmc_valu_$wl.log" 2>&1
  python tools/pmc_valu.py "$(ls "$out"/pmc_valu_$wl/*.db | tail -1)" "$out/pmc_valu_$wl.json" $NST
  rm -rf "$out/trace_$wl" "$out/pmc_fetch_$wl" "$out/pmc_write_$wl" "$out/pmc_valu_$wl"
done
# the default bench line LAST, quoting the counter summaries just made (same kernel sources: `roofline.traffic` is live)
python bench.py --profiles "$out" 2> "$out/bench_stderr.log" | tail -1 > "$out/${tag}_bench_n1.json"
# the edit path, the incremental remesh and the many-object frame under the kernel trace (their own timing lines + kernel tables)
bash tools/prof_edit.sh $tag > "$out/prof_edit.log" 2>&1
bash tools/prof_sync.sh $tag > "$out/prof_sync.log" 2>&1
bash tools/prof_many.sh $tag > "$out/prof_many.log" 2>&1
python3 tools/time_pile.py > "$out/${tag}_time_pile.log" 2>&1
grep -h "C call\|edit enqueue\|^frame\|step_many" "$out/time_edit.log" "$out/time_sync.log" "$out/time_many.log"
head -14 "$out/${tag}_kernel_stats_step_only_headline.csv"
head -14 "$out/${tag}_kernel_stats_step_only_dense.csv"
