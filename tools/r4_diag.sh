#!/bin/bash
# round-4 diagnostic pass on one box: workgroup timelines of the mesher / derive sweep / evaluator on the all-surface grid (trace build:
# build/trace_lib.so) and the SQ wait-state counters of the step's kernels. usage: tools/r4_diag.sh <tag> [headline|dense]
set -u
tag=${1:-d}
wl=${2:-dense}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
if [ -f build/trace_lib.so ]; then
  for st in remesh derive sample; do
    echo "== wg_trace $st $wl" >> "$out/wg_trace.log"
    if [ $wl = dense ]; then A=dense; else A=""; fi
    IMPACT_VOXEL_HIP_LIB=$PWD/build/trace_lib.so python tools/wg_trace.py $st $A >> "$out/wg_trace.log" 2>&1
  done
fi
if [ $wl = dense ]; then export IVX_DIAG_ARGS="--workload dense"; fi
bash tools/pmc_diag.sh $tag/pmc \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" \
  > "$out/pmc_diag.log" 2>&1
cat "$out/wg_trace.log" "$out/pmc_diag.log"
