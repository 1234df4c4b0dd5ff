#!/bin/bash
# the mesher's launch time over N fresh processes on one box (two modes: DESIGN.md section 6 (h)). usage: tools/emit_modes.sh [N] [extra env]
n=${1:-8}
for i in $(seq 1 $n); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pile --plain --workload dense 2>/dev/null | tail -1 > /tmp/em.json
  python - <<'P'
import json
d = json.load(open("/tmp/em.json"))
print("ms/step", round(d["ms_per_step"], 4), "emit", d["stage_ms"]["emit"], "derive", d["stage_ms"]["derive"], "sample", d["stage_ms"]["sdf_sample"])
P
done
