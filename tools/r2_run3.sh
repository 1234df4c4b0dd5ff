#!/bin/bash
set -u
out=$PWD/gpurun_out/r2h
mkdir -p "$out"
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > "$out/r2h_pytest_gpu.log"
tail -3 "$out/r2h_pytest_gpu.log"
IVX_FUZZ_SEEDS=100:260 python -m pytest tests/test_gpu_random_mix.py tests/test_gpu_edit_sequences.py tests/test_gpu_mutual_sequences.py -q -m gpu --tb=line 2>&1 | grep -v "^\\." | cut -c1-300 | tail -20 > "$out/r2h_fuzz_100_260.log"
tail -3 "$out/r2h_fuzz_100_260.log"
python bench.py --no-cpu-baseline 2> "$out/bench_stderr.log" | tail -1 > "$out/r2h_bench.json"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2h/r2h_bench.json'))
print(d['ms_per_step'], d['stage_ms']); print('edit',d['edit']); print('dense', d['dense']['ms_per_step'], d['dense']['stage_ms'])
PY
