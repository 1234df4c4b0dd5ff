#!/bin/bash
# quick GPU check of a kernel change: the parity tests named (default: the mesher's), then the step-only bench on both workloads
# usage: tools/r3_quick.sh <tag> [pytest -k expression | all | none]
set -u
tag=${1:-q}
sel=${2:-"parity or mesh or golden or selftests or edit"}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
if [ "$sel" = "all" ]; then
  python -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log"
elif [ "$sel" != "none" ]; then
  python -m pytest tests -m gpu -x -q -k "$sel" > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log"
fi
for wl in headline dense; do
  if [ $wl = dense ]; then W="--workload dense"; else W=""; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pile --plain $W 2> "$out/bench_$wl.err" | tail -1 > "$out/bench_$wl.json"
  python - "$out/bench_$wl.json" $wl <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms/step", round(d["ms_per_step"], 4), "stage_ms", {k: v for k, v in d.get("stage_ms", {}).items() if v})
P
done
