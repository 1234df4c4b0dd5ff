#!/usr/bin/env python3
"""Time the rigid-body step of the config-4 pile (resident bodies + contacts) on one GPU and the oracle on the host."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from impact_amd import capi, scenes  # noqa: E402
from impact_amd.physics import PhysicsWorld  # noqa: E402
from impact_amd.voxel import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = Context(0)
bodies, contacts = scenes.sphere_pile_scene(n)
w = PhysicsWorld(ctx)
if "--groups" in sys.argv:
    w.set_solver_groups(int(sys.argv[sys.argv.index("--groups") + 1]))
w.set_bodies(bodies)
t0 = time.perf_counter()
w.prepare_constraints(contacts)
t1 = time.perf_counter()
for _ in range(3):
    r = w.step(0.005)
acc = np.zeros(5)
t2 = time.perf_counter()
K = 20
for _ in range(K):
    r = w.step(0.005)
    acc += r["stage_ms"]
t3 = time.perf_counter()
print(f"n={n} bodies={len(bodies)} contacts={len(contacts)} levels={r['n_levels']} set_contacts(host)={1e3 * (t1 - t0):.2f} ms "
      f"step wall={1e3 * (t3 - t2) / K:.3f} ms", {k: round(float(v) / K, 4) for k, v in zip(capi.PHYSICS_STAGE_NAMES, acc)}, w.solver_info())
if "--oracle" in sys.argv:
    import oracle_lib as ol

    o = ol.OraclePhysics(bodies, config=(8, 0.4, 3, 0.2))
    o.step(contacts, 0.005)
    t = time.perf_counter()
    for _ in range(5):
        o.step(contacts, 0.005)
    print(f"oracle step {1e3 * (time.perf_counter() - t) / 5:.2f} ms (1 thread)")
w.close()
ctx.close()
