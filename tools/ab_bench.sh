#!/bin/bash
# A/B of builds of the library on ONE box (boxes differ by a few per cent): every build/lib_*.so, step-only bench, both workloads,
# interleaved twice. usage: tools/ab_bench.sh [dense|headline|both]
set -u
which=${1:-both}
for rep in 1 2; do
  for v in $(ls build/lib_*.so | sed 's/.*lib_//; s/\.so//'); do
    for wl in headline dense; do
      if [ "$which" != both ] && [ "$which" != $wl ]; then continue; fi
      if [ $wl = dense ]; then W="--workload dense"; else W=""; fi
      IMPACT_VOXEL_HIP_LIB=$PWD/build/lib_$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pile --plain $W 2>/dev/null | tail -1 > /tmp/ab.json
      python - $v $wl <<'P'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], sys.argv[2], "ms/step", round(d["ms_per_step"], 4), {k: v for k, v in d.get("stage_ms", {}).items() if v})
P
    done
  done
done
