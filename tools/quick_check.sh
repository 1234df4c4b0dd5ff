#!/bin/bash
# quick GPU check used while tuning: the plain bench lines of both 512^3 workloads, then the whole GPU test suite
set -u
out=$PWD/gpurun_out/r2m
mkdir -p "$out"
python bench.py --plain --steps 200 2>/dev/null | tail -1 > "$out/plain.json"
python bench.py --plain --steps 100 --workload dense 2>/dev/null | tail -1 > "$out/plain_dense.json"
python - <<'PY'
import json
for f in ('plain','plain_dense'):
    d=json.load(open(f'gpurun_out/r2m/{f}.json')); print(f, round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms'].items() if v})
PY
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > "$out/pytest_gpu.log"; cat "$out/pytest_gpu.log"
