#!/bin/bash
# developer experiment: the all-surface step's mesher launch with the mesh arrays carved from one block at different relative offsets
# (IVX_MESH_ARENA / IVX_MESH_PHASE, ivx_api.hip ensure_mesh_block), three fresh processes per setting. usage (GPU box): tools/mesh_phase_sweep.sh <tag>
tag=${1:-phase}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-pile --plain --workload dense"
run() {  # label, env...
  label=$1; shift
  for i in 1 2 3; do
    env "$@" python3 bench.py $ARGS 2>/dev/null | tail -1 > /tmp/ph.json
    python3 - "$label" <<'P'
import json, sys
d = json.load(open("/tmp/ph.json"))
print(f"{sys.argv[1]}: ms/step {d['ms_per_step']:.4f} emit {d['stage_ms']['emit']:.4f}")
P
  done
}
{
run "separate allocations (default)" IVX_NOP=1
for ph in 0 256 1024 4096 16384 65536 262144 1048576 1310720 3145728; do run "hipMalloc block, phase $ph" IVX_MESH_ARENA=1 IVX_MESH_PHASE=$ph; done
for ph in 0 4096 65536 1048576 1310720 5242880; do run "contiguous block, phase $ph" IVX_MESH_ARENA=2 IVX_MESH_PHASE=$ph; done
} 2>&1 | tee "$out/mesh_phase_sweep.log"
