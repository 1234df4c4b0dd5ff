#!/bin/bash
# kernel times of the edit path (tools/time_edit.py under the kernel trace). usage (GPU box): tools/prof_edit.sh <tag>
tag=${1:-pe}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace -- python3 tools/time_edit.py > "$out/time_edit.log" 2>&1
python tools/rocpd_stats.py "$(ls "$out"/trace/*.db | tail -1)" > "$out/kernel_stats_edit.csv"
rm -rf "$out/trace"
cat "$out/time_edit.log" | tail -2
head -30 "$out/kernel_stats_edit.csv"
