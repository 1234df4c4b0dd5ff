#!/usr/bin/env python3
"""Developer tool: ivx_world_set_contacts + step for a contact set that repeats frame after frame, host times per call
(IVX_WORLD_TRACE=1 in the environment prints when a frame leaves the one-pass path). usage: time_world_frames.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from impact_amd import capi, scenes
from impact_amd.physics import PhysicsWorld, uniform_sphere_body
from impact_amd.voxel import Context

ctx = Context(0)
n, per = 81, 48
bodies = np.array([uniform_sphere_body(8.0, 1.0, (20.0 * (k % 9), 0.0, 20.0 * (k // 9))) for k in range(n)])
bodies["total_force"][:, 1] = np.float32(-9.81) * bodies["mass"]
ground = np.zeros(1, dtype=capi.KINEMATIC_BODY_DTYPE)
ground["orientation"] = (0, 0, 0, 1)
ground["angular_axis"] = (0, 1, 0)
cs = np.zeros(n * per, dtype=capi.CONTACT_DTYPE)
rng = np.random.default_rng(1)
for k in range(n):
    for j in range(per):
        c = cs[k * per + j]
        c["id"] = scenes.contact_id(1000 + k, 7, j)
        c["body_a"], c["body_b"] = k, 0x80000000
        c["position"] = (20.0 * (k % 9) + rng.uniform(-3, 3), -8.0, 20.0 * (k // 9) + rng.uniform(-3, 3))
        c["normal"] = (0, 1, 0)
        c["depth"] = 0.05
        c["restitution"], c["static_friction"], c["dynamic_friction"] = 0.2, 0.7, 0.5
        c["flags"] = 1 if j == 0 else 0
for groups in (0, 255):
    w = PhysicsWorld(ctx)
    w.set_bodies(bodies, ground)
    w.set_solver_groups(groups)
    ts, tw = [], []
    for f in range(8):
        ctx.synchronize()
        t0 = time.perf_counter()
        w.prepare_constraints(cs)
        t1 = time.perf_counter()
        w.step_enqueue(0.005)
        t2 = time.perf_counter()
        ctx.synchronize()
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    print(f"groups {groups}:", w.solver_info()["kernel"], [tuple(round(1e3 * x, 3) for x in t) for t in ts])
    w.close()
ctx.close()
