#!/usr/bin/env python3
"""Developer tool: does the mesher's mode change INSIDE a process? One all-surface object, stepped for a minute; the emit launch's median over
every batch of 20 steps with the wall time, then the same for a second object made afterwards. usage: emit_watch.py [seconds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from impact_amd import capi, scenes
from impact_amd.voxel import Context

ctx = Context(0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
for which in range(2):
    gen, obj = bench.make_object(ctx, scenes.plates_scene(32))
    bench.time_steps(ctx, obj, capi.STAGE_ALL, 10, 5)
    t0 = time.perf_counter()
    line = []
    while time.perf_counter() - t0 < secs / 2:
        bench.time_steps(ctx, obj, capi.STAGE_ALL, 20, 0)
        line.append(round(float(np.median(bench.time_steps.last_samples[:, 4])), 3))
        time.sleep(0.25)
    print(f"object {which}: emit medians over {secs / 2:.0f} s:", line)
    obj.close()
