#!/bin/bash
# round 5: the chain-stationary solve on the GPU box — parity tests, then the pile timed on each kernel
set -o pipefail
mkdir -p gpurun_out/r5s
timeout 900 python3 -m pytest tests/test_gpu_physics.py tests/test_gpu_physics_random.py -x -q > gpurun_out/r5s/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5s/pytest.log
tail -3 gpurun_out/r5s/pytest.log
for nap in 0 10 20 30 50; do echo "nap $nap"; IVX_SOLVER_NAP=$nap timeout 300 python3 tools/time_pile.py 16 2>&1 | tail -1 | cut -c1-260; done | tee gpurun_out/r5s/pile_naps.log
timeout 300 python3 tools/time_pile.py 16 --groups 8 2>&1 | tail -1 | cut -c1-260 | tee gpurun_out/r5s/pile_mg8.log
IVX_SOLVER_TRACE=gpurun_out/r5s/trace_nap20.bin timeout 300 python3 tools/time_pile.py 16 2>&1 | tail -1 | cut -c1-200
IVX_SOLVER_NAP=0 IVX_SOLVER_TRACE=gpurun_out/r5s/trace_nap0.bin timeout 300 python3 tools/time_pile.py 16 2>&1 | tail -1 | cut -c1-200
python3 tools/solver_trace.py gpurun_out/r5s/trace_nap20.bin | tee gpurun_out/r5s/trace_nap20.txt
python3 tools/solver_trace.py gpurun_out/r5s/trace_nap0.bin | tee gpurun_out/r5s/trace_nap0.txt
