#!/bin/bash
# the fracture event under the kernel trace: tools/time_cut.py's timing lines + the per-kernel table. usage (GPU box): tools/prof_cut.sh <tag>
tag=${1:-cut}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 tools/time_cut.py 4 --oracle > "$out/time_cut.log" 2>&1
IVX_MANY_TRACE=1 python3 tools/time_cut.py 2 > "$out/time_cut_phases.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace -- python3 tools/time_cut.py 3 > "$out/time_cut_prof.log" 2>&1
python3 tools/rocpd_stats.py "$(ls "$out"/trace/*.db | tail -1)" > "$out/kernel_stats_cut.csv"
rm -rf "$out/trace"
tail -12 "$out/time_cut.log"
grep "ivx many" "$out/time_cut_phases.log" | tail -20
head -20 "$out/kernel_stats_cut.csv" | cut -c1-160
