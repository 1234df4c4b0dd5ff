#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, the default bench line, the rocprofv3 kernel trace of the same bench
# command and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no other trace domains). Summaries land in
# gpurun_out/<tag>/ and are copied into profiles/<round>/ by hand afterwards.
# usage: tools/profile_round.sh <tag>
set -u
tag=${1:-run}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > "$out/${tag}_pytest_gpu.log" 2>&1
tail -2 "$out/${tag}_pytest_gpu.log"
python bench.py 2> "$out/bench_stderr.log" | tail -1 > "$out/${tag}_bench_n1.json"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$out/trace_stdout.log" 2>&1
python tools/rocpd_stats.py "$(ls "$out"/trace/*.db | tail -1)" > "$out/${tag}_kernel_stats_bench_steps10.csv"
# the step workload alone (no pile / edit / collide legs): every launch of a step kernel is a step launch, so the averages
# here are the ones bench.py's per-stage HIP-event times must agree with
rocprofv3 --kernel-trace --stats -d "$out/trace_step" -o trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pile > "$out/${tag}_bench_step_only.json" 2> "$out/trace_step_stderr.log"
python tools/rocpd_stats.py "$(ls "$out"/trace_step/*.db | tail -1)" > "$out/${tag}_kernel_stats_step_only.csv"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pile > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pile > "$out/pmc_write.log" 2>&1
python tools/pmc_traffic.py "$(ls "$out"/pmc_fetch/*.db | tail -1)" "$(ls "$out"/pmc_write/*.db | tail -1)" "$out/pmc_traffic.json"
rm -rf "$out/trace" "$out/trace_step" "$out/pmc_fetch" "$out/pmc_write"
head -12 "$out/${tag}_kernel_stats_bench_steps10.csv"
