#!/usr/bin/env python3
"""Time the stages of ONE slab of the N-rank weak-scaling workload on a single GPU (no neighbours: ghost
layers empty). Shows how the per-rank stage times change with the size of the N-body SDF program.
usage: time_slab.py N RANK [scale]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from impact_amd import capi, scenes  # noqa: E402
from impact_amd.distributed import slab_ranges  # noqa: E402
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject  # noqa: E402

n, rank = int(sys.argv[1]), int(sys.argv[2])
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 2.05
ctx = Context(0)
gen = SDFVoxelGenerator(1.0, scenes.asteroid_row_scene(n, scale), 0)
cc = gen.chunk_counts()
x0, x1 = slab_ranges(cc[0], n)[rank]
obj = VoxelObject(ctx, (x1 - x0, cc[1], cc[2]), 1.0, x0, cc[0])
obj.set_sdf_program(gen)
obj.set_densities(np.ones(256, dtype=np.float32))
for _ in range(3):
    obj.step(capi.STAGE_ALL)
acc = np.zeros(capi.N_TIMED_STAGES)
for _ in range(10):
    acc += obj.step(capi.STAGE_ALL)["stage_ms"]
print(f"N={n} rank={rank} nodes={len(gen.sdf_generator.nodes)} stack={gen.sdf_generator.required_forward_stack_size} chunks={obj.n_chunks}",
      {k: round(float(v) / 10, 4) for k, v in zip(capi.STAGE_NAMES, acc)}, "total", round(float(acc.sum()) / 10, 3))
obj.close()
ctx.close()
