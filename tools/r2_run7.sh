#!/bin/bash
set -u
out=$PWD/gpurun_out/r2g
mkdir -p "$out"
python -m pytest tests/test_gpu_slabs.py -m gpu -q -x 2>&1 | tail -8 > "$out/r2g_pytest_slabs.log"; tail -4 "$out/r2g_pytest_slabs.log"
python bench.py --no-cpu-baseline --no-pile 2>/dev/null | tail -1 > "$out/r2g_plain.json"
IVX_BENCH_FORCE_SLABS=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-pile 2> "$out/slab_stderr.log" | tail -1 > "$out/r2g_slab_world1_rccl.json"
python - <<'PY'
import json
for f in ('r2g_plain.json','r2g_slab_world1_rccl.json'):
    try:
        d=json.load(open('gpurun_out/r2g/'+f)); print(f, round(d['ms_per_step'],4), d['config']['parallelism'][:90], d['config']['triangles'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -5 "$out/slab_stderr.log"
