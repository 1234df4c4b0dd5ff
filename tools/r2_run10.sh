#!/bin/bash
set -u
out=$PWD/gpurun_out/r2n
mkdir -p "$out"
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > "$out/pytest_gpu.log"; cat "$out/pytest_gpu.log"
python bench.py 2>"$out/bench_stderr.log" | tail -1 > "$out/bench.json"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2n/bench.json'))
print('step', d['ms_per_step'], d['value'], d.get('stage_ms'))
for k in ('edit','pile','frame','config2','config3','collide','dense','remesh'):
    if k in d: print(k, json.dumps(d[k])[:400])
PY
