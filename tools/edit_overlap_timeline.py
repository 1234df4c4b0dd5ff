#!/usr/bin/env python3
"""Developer tool: the overlapped edit + sync (edit enqueue -> sync enqueue(None) -> edit collect -> sync collect) a few times, host times of
the four calls; under tools/timeline.sh the kernel list of the last repetition. usage: edit_overlap_timeline.py [reps]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import bench
from impact_amd import capi, scenes
from impact_amd.voxel import Context, VoxelObjectMesh

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scale = 2.05
ctx = Context(0)
_, obj = bench.make_object(ctx, scenes.asteroid_scene(scale))
mesh = VoxelObjectMesh(obj)
obj.set_early_mesh_needs(True)
for rep in range(reps):
    obj.step(capi.STAGE_ALL)
    c = np.array([0.5 * (a + b) for a, b in obj.update_occupied_voxel_ranges()], dtype=np.float32) + bench.EDIT_OFFSET * np.float32(scale)
    mesh.sync_with_voxel_object(np.zeros(obj.n_chunks, dtype=np.uint8))
    ctx.synchronize()
    time.sleep(0.002)
    t0 = time.perf_counter()
    obj.absorb_sphere_enqueue(c, bench.EDIT_RADIUS * scale + 2.0, bench.EDIT_RADIUS * scale)
    t1 = time.perf_counter()
    mesh.sync_enqueue(None)
    t2 = time.perf_counter()
    r = obj.absorb_collect(want_invalidated=True)
    t3 = time.perf_counter()
    mesh.sync_collect()
    t4 = time.perf_counter()
    time.sleep(0.002)
    print(f"edit enqueue {1e6 * (t1 - t0):.1f} us, sync enqueue {1e6 * (t2 - t1):.1f}, edit collect {1e6 * (t3 - t2):.1f}, sync collect {1e6 * (t4 - t3):.1f}, total {1e6 * (t4 - t0):.1f}")
