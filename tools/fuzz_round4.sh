#!/bin/bash
# Round-4 sweep of the randomized parity tests on the GPU box (one pytest process per group; the committed seeds stay the default of a plain
# `pytest -m gpu`): tools/fuzz_round4.sh <first seed> <one past the last seed> [tag]
set -u
export IVX_FUZZ_SEEDS="${1:-100}:${2:-140}"
tag=${3:-r4_fuzz}
mkdir -p gpurun_out
log=gpurun_out/${tag}_${1:-100}_${2:-140}.log
: > "$log"
for grp in "tests/test_gpu_random_sdf.py" "tests/test_gpu_random_mix.py tests/test_gpu_edit_sequences.py tests/test_gpu_mutual_sequences.py" \
           "tests/test_gpu_clip.py tests/test_gpu_slabs.py tests/test_gpu_contacts.py" "tests/test_gpu_physics_random.py"; do
  echo "== $grp (seeds $IVX_FUZZ_SEEDS)" >> "$log"
  python -m pytest $grp -q -m gpu --tb=line 2>&1 | grep -v "^\\.\\|^$\\|RCCL\\|HIP version\\|ROCm version\\|Hostname\\|Librccl" | cut -c1-300 | tail -12 >> "$log"
done
cat "$log"
