#!/usr/bin/env python3
"""Host-side cost of the slab protocol: wall time per step of (a) the plain single-GPU step, (b) the same grid driven through
SlabStepper.phases as a one-slab "decomposition" in one process (all protocol calls, torch stream context, record readback, no
neighbour traffic), (c) slab `rank` of bench.py's N-rank workload alone (no neighbour traffic). usage: time_protocol.py [N RANK]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from impact_amd import capi, scenes  # noqa: E402
from impact_amd.distributed import SlabStepper, run_slabs_in_process  # noqa: E402
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject  # noqa: E402

ctx = Context(0)
dens = np.ones(256, dtype=np.float32)


def wall(fn, n=30):
    for _ in range(5):
        fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(2.05), 0)
obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
obj.set_sdf_program(gen)
obj.set_densities(dens)
print(f"plain step: {wall(lambda: obj.step(capi.STAGE_ALL)):.3f} ms")
obj.close()
st = SlabStepper(ctx, scenes.asteroid_scene(2.05), dens, 0, 1, torch)
print(f"one-slab protocol in process: {wall(lambda: run_slabs_in_process([st])):.3f} ms")
st.close()
if len(sys.argv) > 2:
    n, r = int(sys.argv[1]), int(sys.argv[2])
    st = SlabStepper(ctx, scenes.asteroid_scene(2.05 * n ** (1.0 / 3.0)), dens, r, n, torch)  # bench.py's N-rank workload
    st.has_lo = st.has_hi = False  # no neighbours in this process: empty ghost layers
    st.rank = 0  # (its record is the only one gathered here)
    print(f"slab {r} of {n} alone through the protocol: {wall(lambda: run_slabs_in_process([st])):.3f} ms")
    st.close()
ctx.close()
