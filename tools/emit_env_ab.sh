#!/bin/bash
# developer experiment: the mesher's mode under runtime settings, interleaved (A B C D A B C D ...) because the mode comes in streaks
tag=${1:-envab}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-pile --plain --workload dense"
one() {
  label=$1; shift
  env "$@" python3 bench.py $ARGS 2>/dev/null | tail -1 > /tmp/ab.json
  python3 - "$label" <<'P'
import json, sys
try:
    d = json.load(open("/tmp/ab.json"))
    print(f"{sys.argv[1]}: emit {d['stage_ms']['emit']:.4f} step {d['ms_per_step']:.4f}")
except Exception as e:
    print(sys.argv[1], "failed", e)
P
}
{
for r in 1 2 3 4 5 6; do
  one "default" IVX_NOP=1
  one "GPU_MAX_HW_QUEUES=1" GPU_MAX_HW_QUEUES=1
  one "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
  one "AMD_DIRECT_DISPATCH=0" AMD_DIRECT_DISPATCH=0
  one "HSA_ENABLE_SDMA=0" HSA_ENABLE_SDMA=0
done
} 2>&1 | tee "$out/emit_env_ab.log"
