#!/usr/bin/env python3
"""Developer tool: the all-surface step as bench.py's `dense` leg times it, before and after the oracle has run in the process
(one thread, then OpenMP): per-step wall times."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from impact_amd import capi, scenes
from impact_amd.voxel import Context

ctx = Context(0)
gen, obj = bench.make_object(ctx, scenes.plates_scene(32))


def run(tag):
    for _ in range(2):
        obj.step(capi.STAGE_ALL)
    ctx.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        obj.step_enqueue(capi.STAGE_ALL)
        t1 = time.perf_counter()
        r = obj.step_collect()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    ts = 1e3 * np.array(ts)
    print(f"{tag}: enqueue {np.median(ts[:, 0]):.3f} ms, collect {np.median(ts[:, 1]):.3f} ms (max {ts[:, 0].max():.3f} / {ts[:, 1].max():.3f}); stage sum {np.sum(r['stage_ms']):.3f}")


run("fresh")
g2 = scenes.asteroid_scene(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
o, m, t = bench.cpu_voxel_step(g2, 1)
run("after the oracle, one thread")
o2, m2, t2 = bench.cpu_voxel_step(g2, 16)
run("after the oracle, 16 threads")
time.sleep(2.0)
run("2 s later")
obj.close()
gen, obj = bench.make_object(ctx, scenes.plates_scene(32))
run("a new object")
res, ms, st = bench.time_steps(ctx, obj, capi.STAGE_ALL, 50, 2)
print("time_steps:", ms, st.sum())
del o, m, o2, m2
run("oracle objects dropped")
obj.close()
gen, obj = bench.make_object(ctx, scenes.plates_scene(32))
run("a new object again")
