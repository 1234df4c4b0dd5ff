#!/usr/bin/env python3
"""Developer tool: the bench's edit (absorbing sphere into the headline body + incremental remesh), a few repetitions, host times printed;
run under tools/timeline.sh to get the ordered kernel list of the last repetition. usage: edit_timeline.py [reps]"""
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
for rep in range(reps):
    obj.step(capi.STAGE_ALL)
    c = np.array([0.5 * (a + b) for a, b in obj.update_occupied_voxel_ranges()], dtype=np.float32) + bench.EDIT_OFFSET * np.float32(scale)
    mesh.sync_with_voxel_object(np.zeros(obj.n_chunks, dtype=np.uint8))
    ctx.synchronize()
    time.sleep(0.002)  # (a visible gap in the kernel trace ahead of the edit)
    t0 = time.perf_counter()
    r = obj.absorb_sphere(c, bench.EDIT_RADIUS * scale + 2.0, bench.EDIT_RADIUS * scale, want_invalidated=True)
    t1 = time.perf_counter()
    mesh.sync_with_voxel_object(r["invalidated"])
    t2 = time.perf_counter()
    ctx.synchronize()
    time.sleep(0.002)
    print(f"edit {1e3 * (t1 - t0):.4f} ms, sync {1e3 * (t2 - t1):.4f} ms, touched {r['touched_chunks']}, invalidated {int(r['invalidated'].sum())}")
