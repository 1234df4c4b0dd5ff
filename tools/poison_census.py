#!/usr/bin/env python3
"""Developer tool: how often the sampler's pre-pass falls back to the full program because a saturation-class drop was followed by a
combination behind the 14-position test (`poisoned`, sdf_sample.hip) — over the random SDF programs of tests/test_gpu_random_sdf.py.
usage: poison_census.py [first seed] [one past the last]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from impact_amd import capi  # noqa: E402
from impact_amd.sdf_graph import SDFGraph  # noqa: E402
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject  # noqa: E402
from test_gpu_random_sdf import random_tree  # noqa: E402

a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 1300)
ctx = Context(0)
lib = capi.lib()
lib.ivx_grid_device_ptr.restype = C.c_void_p
hip = C.CDLL("libamdhip64.so")
tot_chunks = tot_over = progs_with = progs = 0
for seed in range(a, b):
    rng = np.random.default_rng(seed)
    g = SDFGraph()
    random_tree(g, rng, int(rng.integers(1, 5)))
    extent = [1.0, 0.5, 0.25, 2.0][seed % 4]
    gen = SDFVoxelGenerator(extent, g, 0)
    if min(gen.chunk_counts()) == 0:
        continue
    obj = VoxelObject(ctx, gen.chunk_counts(), extent)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    obj.step(capi.STAGE_SAMPLE)
    n = obj.n_chunks
    lens = np.zeros(n, dtype=np.uint32)
    hip.hipDeviceSynchronize()
    assert hip.hipMemcpy(lens.ctypes.data_as(C.c_void_p), C.c_void_p(lib.ivx_grid_device_ptr(obj.h, 7)), lens.nbytes, 2) == 0
    over = int(np.count_nonzero(lens == 0xFFFFFFFF))
    progs += 1
    tot_chunks += n
    tot_over += over
    progs_with += 1 if over else 0
    if over:
        print(f"seed {seed}: {over} of {n} chunks on the full program ({len(gen.sdf_generator.nodes)} nodes)")
    obj.close()
print(f"{progs} programs, {tot_chunks} chunks: {tot_over} chunks of {progs_with} programs evaluated on the full program")
