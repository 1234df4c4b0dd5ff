#!/usr/bin/env python3
"""Which role of k_step_post1 costs what (developer tool): the all-surface 512^3 workload stepped with one post stage at a time —
post1 then holds only that stage's role(s). usage: time_post1_roles.py [dense|headline]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impact_amd import capi, scenes  # noqa: E402
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject  # noqa: E402


def main():
    dense = len(sys.argv) < 2 or sys.argv[1] == "dense"
    ctx = Context(0)
    gen = SDFVoxelGenerator(1.0, scenes.plates_scene(32) if dense else scenes.asteroid_scene(2.05), 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    for _ in range(3):
        obj.step(capi.STAGE_ALL)
    for name, st in (("all", capi.STAGE_ALL), ("remesh", capi.STAGE_REMESH), ("regions", capi.STAGE_REGIONS), ("occupied", capi.STAGE_OCCUPIED),
                     ("inertia", capi.STAGE_INERTIA)):
        acc = np.zeros(capi.N_TIMED_STAGES)
        n = 10
        for _ in range(2):
            obj.step(st)
        for _ in range(n):
            acc += obj.step(st)["stage_ms"]
        print(name, {capi.STAGE_NAMES[i]: round(float(acc[i] / n), 4) for i in range(capi.N_TIMED_STAGES) if acc[i] > 0})
    obj.close()
    ctx.close()


if __name__ == "__main__":
    main()
