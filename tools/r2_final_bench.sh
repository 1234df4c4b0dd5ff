#!/bin/bash
# the bench lines that go into profiles/round2 (run after the PMC summaries of the same build are in place: bench.py reads them)
out=$PWD/gpurun_out/${1:-r2q}
mkdir -p "$out"
python bench.py 2> "$out/bench_stderr.log" | tail -1 > "$out/bench_n1.json"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pile --plain 2>/dev/null | tail -1 > "$out/bench_step_only_headline.json"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pile --plain --workload dense 2>/dev/null | tail -1 > "$out/bench_step_only_dense.json"
python - "$out" <<'PY'
import json, sys
for f in ('bench_n1', 'bench_step_only_headline', 'bench_step_only_dense'):
    d = json.load(open(f'{sys.argv[1]}/{f}.json'))
    r = d['roofline']
    print(f, round(d['ms_per_step'], 4), 'ms', round(d['value'] / 1e9, 1), 'G voxels/s |', r['stage'], 'frac', round(r['frac'], 3), 'counter_frac', r.get('counter_frac'), 'traffic', r.get('traffic'))
PY
