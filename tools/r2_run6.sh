#!/bin/bash
for lib in libimpact_voxel_hip.so libimpact_voxel_hip_v2.so; do
  for wl in asteroid dense; do
    echo "== $lib $wl"
    IMPACT_VOXEL_HIP_LIB=$PWD/impact_amd/lib/$lib python bench.py --no-cpu-baseline --no-pile --workload $wl --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), {k:v for k,v in d['stage_ms'].items() if v}, 'remesh', round(d['remesh_ms'],4))"
  done
done
