#!/bin/bash
# kernel times of edit + incremental remesh (tools/time_sync.py under the kernel trace). usage (GPU box): tools/prof_sync.sh <tag>
tag=${1:-ps}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 tools/time_sync.py > "$out/time_sync.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace -- python3 tools/time_sync.py > "$out/time_sync_prof.log" 2>&1
python tools/rocpd_stats.py "$(ls "$out"/trace/*.db | tail -1)" > "$out/kernel_stats_sync.csv"
rm -rf "$out/trace"
grep "edit enqueue" "$out/time_sync.log"
cut -c1-120 "$out/kernel_stats_sync.csv" | head -24
