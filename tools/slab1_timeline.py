#!/usr/bin/env python3
"""Developer tool: the slab protocol at world size 1 (in-process communicator) on the headline body, a few steps; under tools/timeline.sh the
ordered kernel list of the last ones. usage: slab1_timeline.py [steps] (SLAB_SAMPLE_AHEAD=0 turns the pre-pass a step ahead off)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from impact_amd import scenes
from impact_amd.distributed import NativeComm, NativeSlabStepper, NativeStepGroup
from impact_amd.voxel import Context

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = Context(0)
comm = NativeComm(ctx, 1, local=True)
st = NativeSlabStepper(ctx, comm, scenes.asteroid_scene(2.05), np.ones(256, dtype=np.float32), 0, sample_ahead=os.environ.get("SLAB_SAMPLE_AHEAD", "1") != "0")
group = NativeStepGroup([st])
st.obj.set_stage_timing(0)
for _ in range(3):
    group.step()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    group.step()
print(f"{1e3 * (time.perf_counter() - t0) / steps:.4f} ms per step")
