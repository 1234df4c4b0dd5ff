#!/usr/bin/env python3
"""Where does the wall time of a bench step go beyond the kernels? (N=1 workload)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from impact_amd import capi, scenes
from impact_amd.physics import PhysicsWorld, uniform_sphere_body
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject

ctx = Context(0)
gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(2.05), 0)
obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
obj.set_sdf_program(gen)
obj.set_densities(np.ones(256, dtype=np.float32))
w = PhysicsWorld(ctx)
w.set_bodies(np.array([uniform_sphere_body(100.0, 1.0, (0, 0, 0), (0.1, 0, 0))]))
w.prepare_constraints(np.zeros(0, dtype=capi.CONTACT_DTYPE))
for _ in range(5):
    obj.step(); w.step(0.005)
K = 50
def T(f):
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(K): f()
    ctx.synchronize(); return 1e3 * (time.perf_counter() - t) / K
ks = np.zeros(10)
def full():
    global ks
    ks += obj.step()["stage_ms"]
print("voxel step wall %.3f ms" % T(full), "kernel sum %.3f" % (ks.sum() / K))
print("enqueue only %.3f ms" % T(lambda: obj.step_enqueue(capi.STAGE_ALL)))
obj.step_collect()
print("body step wall %.3f ms" % T(lambda: w.step(0.005)))
print("empty sync %.4f ms" % T(lambda: ctx.synchronize()))
