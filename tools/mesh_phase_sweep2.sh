#!/bin/bash
tag=${1:-phase2}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-pile --plain --workload dense"
run() {
  label=$1; n=$2; shift; shift
  for i in $(seq 1 $n); do
    env "$@" python3 bench.py $ARGS 2>/dev/null | tail -1 > /tmp/ph.json
    python3 - "$label" <<'P'
import json, sys
d = json.load(open("/tmp/ph.json"))
print(f"{sys.argv[1]}: ms/step {d['ms_per_step']:.4f} emit {d['stage_ms']['emit']:.4f}")
P
  done
}
{
run "hipMalloc block, phase 1024" 8 IVX_MESH_ARENA=1 IVX_MESH_PHASE=1024
for ph in 512 768 1280 1536 2048 3072 5120 9216 33792; do run "hipMalloc block, phase $ph" 3 IVX_MESH_ARENA=1 IVX_MESH_PHASE=$ph; done
run "separate allocations (default)" 4 IVX_NOP=1
} 2>&1 | tee "$out/mesh_phase_sweep2.log"
