#!/bin/bash
set -u
out=$PWD/gpurun_out/r2e
mkdir -p "$out"
python -m pytest tests/test_gpu_physics.py tests/test_gpu_physics_random.py tests/test_gpu_contacts.py tests/test_gpu_collide.py -m gpu -q -x 2>&1 | tail -15 > "$out/r2e_pytest_physics.log"
tail -5 "$out/r2e_pytest_physics.log"
for g in 0 1 2 3 4 6 8 12 16; do python tools/time_pile.py 16 --groups $g 2>&1 | tail -1; done > "$out/r2e_time_pile.log"; cat "$out/r2e_time_pile.log"
