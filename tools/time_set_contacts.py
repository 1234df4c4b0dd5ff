#!/usr/bin/env python3
"""ivx_world_set_contacts on a contact set that CHANGES every frame (the path a moving pile takes): the config-4 pile, each frame a tenth of its
manifolds leave and the tenth that left the frame before return. Prints the host time of the call (min / median / max over the frames) and of
set_contacts + step together; IVX_WORLD_TRACE=2 in the environment makes the library print its own laps for every call.
usage: time_set_contacts.py [n] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from impact_amd import scenes  # noqa: E402
from impact_amd.physics import PhysicsWorld  # noqa: E402
from impact_amd.voxel import Context  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    ctx = Context(0)
    bodies, contacts = scenes.sphere_pile_scene(n)
    w = PhysicsWorld(ctx)
    w.set_bodies(bodies)
    w.prepare_constraints(contacts)
    w.step(0.005)
    lists = scenes.pile_churn_frames(contacts, frames)
    t_set, t_all = [], []
    for cs in lists:
        ctx.synchronize()
        t0 = time.perf_counter()
        w.prepare_constraints(cs)
        t1 = time.perf_counter()
        w.step(0.005)
        t2 = time.perf_counter()
        t_set.append(1e3 * (t1 - t0))
        t_all.append(1e3 * (t2 - t0))
    q = lambda a: f"min {min(a):.3f} median {float(np.median(a)):.3f} max {max(a):.3f} ms"  # noqa: E731
    print(f"n={n} contacts={len(contacts)} frames={frames}: set_contacts {q(t_set)} | set_contacts + step {q(t_all)} | {w.solver_info()}")
    w.close()
    ctx.close()


if __name__ == "__main__":
    main()
