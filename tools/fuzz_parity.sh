#!/bin/bash
# Wider sweep of the randomized parity tests on the GPU box: tools/fuzz_parity.sh <first seed> <one past the last seed>
# (one pytest process; the committed seeds stay the default of a plain `pytest -m gpu`)
set -u
export IVX_FUZZ_SEEDS="${1:-100}:${2:-140}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_random_mix.py tests/test_gpu_edit_sequences.py tests/test_gpu_mutual_sequences.py tests/test_gpu_physics_random.py tests/test_gpu_slabs.py tests/test_gpu_contacts.py tests/test_gpu_clip.py -q -m gpu --tb=line 2>&1 | grep -v "^\\." | cut -c1-400 | tail -40 | tee gpurun_out/fuzz_${1:-100}_${2:-140}.log
