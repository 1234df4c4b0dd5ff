#!/usr/bin/env python3
"""Developer tool: where the time of the bench's edit goes — the C call alone (arguments prepared once), the Python wrapper around it,
and a DERIVE|REGIONS step of the same object for comparison."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impact_amd import capi, scenes  # noqa: E402
from impact_amd.capi import ptr  # noqa: E402
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject  # noqa: E402


def main():
    scale = 2.05
    ctx = Context(0)
    gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(scale), 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    dens = np.ones(256, dtype=np.float32)
    obj.set_densities(dens)
    lib = capi.lib()
    out = np.zeros(1, dtype=capi.ABSORB_RESULT_DTYPE)
    by_type = np.zeros(256, dtype=np.uint32)
    inval = np.zeros(obj.n_chunks, dtype=np.uint8)
    t_c, t_py, t_step, t_step0 = [], [], [], []
    for rep in range(8):
        obj.step(capi.STAGE_ALL)
        obj.set_stage_timing(0)
        ctx.synchronize()
        t0 = time.perf_counter()
        obj.step_enqueue(capi.STAGE_DERIVE | capi.STAGE_REGIONS)
        obj.step_collect()
        t1 = time.perf_counter()
        obj.set_stage_timing(0xFFFFFFFF)
        t_step0.append(t1 - t0)
        c = (np.array([0.5 * (a + b) for a, b in obj.update_occupied_voxel_ranges()], dtype=np.float32) + np.array([110.0, 6.0, -4.0], dtype=np.float32) * np.float32(scale))
        ctx.synchronize()
        t0 = time.perf_counter()
        rc = lib.ivx_absorb_sphere(obj.h, ptr(c), C.c_float(15.0 * scale + 2.0), C.c_float(15.0 * scale), ptr(dens), ptr(out), ptr(by_type), ptr(inval))
        t1 = time.perf_counter()
        assert rc == 0
        t_c.append(t1 - t0)
        obj.step(capi.STAGE_ALL)
        ctx.synchronize()
        t0 = time.perf_counter()
        obj.absorb_sphere(c, 15.0 * scale + 2.0, 15.0 * scale, want_invalidated=True)
        t1 = time.perf_counter()
        t_py.append(t1 - t0)
        obj.set_stage_timing(0)
        ctx.synchronize()
        t0 = time.perf_counter()
        obj.step_enqueue(capi.STAGE_DERIVE | capi.STAGE_REGIONS)
        obj.step_collect()
        t1 = time.perf_counter()
        obj.set_stage_timing(0xFFFFFFFF)
        t_step.append(t1 - t0)
    print("C call ms", round(1e3 * float(np.mean(t_c[2:])), 4), " python wrapper ms", round(1e3 * float(np.mean(t_py[2:])), 4),
          " derive+regions step ms", round(1e3 * float(np.mean(t_step[2:])), 4), " the same before any edit ms", round(1e3 * float(np.mean(t_step0[2:])), 4))


if __name__ == "__main__":
    main()
