#!/bin/bash
# A/B of the step's completion wait: doorbell poll (default) against the runtime's blocking wait (IVX_COLLECT_SPIN_US=0)
set -u
out=$PWD/gpurun_out/r2l
mkdir -p "$out"
for rep in 1 2; do
  IVX_COLLECT_SPIN_US=0 python bench.py --plain --steps 200 2>/dev/null | tail -1 > "$out/block_$rep.json"
  python bench.py --plain --steps 200 2>/dev/null | tail -1 > "$out/spin_$rep.json"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2l/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms'].items() if v})
    except Exception as e: print(f,'ERR',e)
PY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_physics.py -m gpu -x -q 2>&1 | tail -3
python tools/prog_stats.py 2>&1 | tail -4
python -m pytest tests/test_gpu_random_sdf.py tests/test_gpu_sample.py -m gpu -x -q 2>&1 | tail -3
