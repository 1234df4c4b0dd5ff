#!/bin/bash
# Developer tool: are the slab tests sensitive to a ghost layer (or the neighbour's face ids) read ahead of its arrival? The overlapped in-process
# mode overwrites the receive buffers before every exchange and moves the messages behind a 150 us delay (slab_comm.cpp, `scribble`,
# k_slow_link); with one of the three waits for an arrival event left out (IVX_DEBUG_SKIP_GHOST_WAIT = 1 the derive sweep's, 2 the mesher
# count's, 3 the face-pair pass's) the native_overlap tests must FAIL, with all in place they pass.
# usage (GPU box): tools/slab_overlap_check.sh <out dir>
out=${1:-gpurun_out/slab_overlap_check}
mkdir -p $out
python -m pytest tests/test_gpu_slabs.py -q -m gpu -k "native_overlap" > $out/with_waits.log 2>&1
echo "all waits in place: exit $? ($(grep -E 'passed|failed' $out/with_waits.log | tail -1 | cut -c1-80))" | tee $out/summary.txt
for w in 1 2 3; do
    IVX_DEBUG_SKIP_GHOST_WAIT=$w python -m pytest tests/test_gpu_slabs.py -q -m gpu -k "native_overlap" > $out/without_wait_$w.log 2>&1
    echo "without wait $w (must fail): exit $? ($(grep -E 'passed|failed' $out/without_wait_$w.log | tail -1 | cut -c1-80))" | tee -a $out/summary.txt
done
