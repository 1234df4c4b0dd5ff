#!/bin/bash
# round 2, run 1: (a) the new deterministic nested-scaling test against the PRE-FIX library (must fail), (b) the same with the fix,
# (c) the 100:500 seed sweep of the random SDF programs with the fix, (d) the whole GPU suite.
set -u
out=$PWD/gpurun_out/r2a
mkdir -p "$out"
IMPACT_VOXEL_HIP_LIB=$PWD/impact_amd/lib/prefix_r1_libimpact_voxel_hip.so python -m pytest tests/test_gpu_random_sdf.py -q -m gpu --tb=line 2>&1 | grep -v "^\\." | cut -c1-300 | tail -30 > "$out/r2a_prefix_lib_random_sdf.log"
python -m pytest tests/test_gpu_random_sdf.py -q -m gpu --tb=line 2>&1 | cut -c1-300 | tail -15 > "$out/r2a_fixed_random_sdf.log"
IVX_FUZZ_SEEDS=100:500 python -m pytest tests/test_gpu_random_sdf.py -q -m gpu --tb=line 2>&1 | grep -v "^\\." | cut -c1-300 | tail -30 > "$out/r2a_fuzz_sdf_100_500.log"
python -m pytest tests -m gpu -q 2>&1 | tail -15 > "$out/r2a_pytest_gpu.log"
tail -3 "$out"/*.log
