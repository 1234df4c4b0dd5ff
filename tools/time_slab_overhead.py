#!/usr/bin/env python3
"""World-1 overhead of the slab protocol: the 512^3 bench workload stepped as a plain grid and as the single slab of an in-process
communicator (same kernels + packing-free protocol + record publish), steps timed back to back; with a world size > 1 all slabs run in
this process on this GPU, one after the other on one stream — time / world is the GPU time ONE rank of that decomposition needs per step
(no link latency in it). usage: time_slab_overhead.py [steps] [world]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from impact_amd import capi, scenes  # noqa: E402
from impact_amd.distributed import NativeComm, NativeSlabStepper, NativeStepGroup  # noqa: E402
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = Context(0)
graph = scenes.asteroid_scene(2.05)
dens = np.ones(256, dtype=np.float32)
gen = SDFVoxelGenerator(1.0, graph, 0)
obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
obj.set_sdf_program(gen)
obj.set_densities(dens)
obj.set_stage_timing(0)
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
comm = NativeComm(ctx, world, local=True)
sts = [NativeSlabStepper(ctx, comm, graph, dens, r) for r in range(world)]
for st in sts:
    st.obj.set_stage_timing(0)
group = NativeStepGroup(sts)


def plain():
    obj.step_enqueue(capi.STAGE_ALL)
    return obj.step_collect()


for name, fn in (("plain", plain), ("slab", group.step), ("plain", plain), ("slab", group.step)):
    for _ in range(20):
        fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    print(name, "ms/step", round(ms, 4), *(("| per rank", round(ms / world, 4)) if name == "slab" and world > 1 else ()))
