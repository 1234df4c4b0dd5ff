#!/usr/bin/env python3
"""Developer tool: bench.py's config-5 leg, the eight x-slabs of the 1024^3 body stepped on one GPU (in-process communicator), a few steps;
under tools/timeline.sh the ordered kernel list of the last step. usage: slab_timeline.py [steps]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from impact_amd import scenes
from impact_amd.distributed import NativeComm, NativeSlabStepper, NativeStepGroup
from impact_amd.voxel import Context

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = Context(0)
graph = scenes.asteroid_scene(4.2)
comm = NativeComm(ctx, 8, local=True)
if os.environ.get("SLAB_LOCAL_COPIES"):
    comm.set_local_copies(int(os.environ["SLAB_LOCAL_COPIES"]))
steppers = [NativeSlabStepper(ctx, comm, graph, np.ones(256, dtype=np.float32), r, sample_ahead=os.environ.get("SLAB_SAMPLE_AHEAD", "1") != "0") for r in range(8)]
group = NativeStepGroup(steppers)
if os.environ.get("SLAB_NO_STAGE_TIMING"):
    for s_ in steppers:
        s_.obj.set_stage_timing(0)
for _ in range(2):
    group.step()
ctx.synchronize()
for _ in range(steps):
    time.sleep(0.003)
    t0 = time.perf_counter()
    group.step()
    print(f"group step {1e3 * (time.perf_counter() - t0):.4f} ms")
