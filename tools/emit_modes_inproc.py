#!/usr/bin/env python3
"""Developer tool: the all-surface step's mesher time for objects created one after the other in ONE process (does the mode depend on where
the buffers land?). usage: emit_modes_inproc.py [rounds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from impact_amd import capi, scenes
from impact_amd.voxel import Context

ctx = Context(0)
keep = []
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    gen, obj = bench.make_object(ctx, scenes.plates_scene(32))
    res, ms, st = bench.time_steps(ctx, obj, capi.STAGE_ALL, 20, 5)
    print(f"object {rnd}: {ms:.4f} ms/step, emit {st[4]:.4f}, derive {st[1]:.4f}, sample {st[0]:.4f}")
    if rnd % 2:
        keep.append(obj)  # every other object stays alive: the next one's buffers land elsewhere
    else:
        obj.close()
