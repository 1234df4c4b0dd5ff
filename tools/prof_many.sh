#!/bin/bash
# kernel trace of the many-object frame (tools/time_many.py). usage (GPU box): tools/prof_many.sh <tag>
tag=${1:-pm}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 tools/time_many.py > "$out/time_many.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace -- python3 tools/time_many.py > "$out/time_many_prof.log" 2>&1
python tools/rocpd_stats.py "$(ls "$out"/trace/*.db | tail -1)" > "$out/kernel_stats_many.csv"
rm -rf "$out/trace"
cat "$out/time_many.log" | tail -9
head -24 "$out/kernel_stats_many.csv" | cut -c1-150
