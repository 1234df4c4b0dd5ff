#!/usr/bin/env python3
"""Developer tool: what the exact numbering (ccl_exact_chunk) meets after the bench's edit — chunks with several regions, their sources
and merge events — and what a regions-only step costs before and after the edit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impact_amd import capi, scenes
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject

scale = 2.05
ctx = Context(0)
gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(scale), 0)
obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
obj.set_sdf_program(gen)
obj.set_densities(np.ones(256, dtype=np.float32))
obj.step(capi.STAGE_ALL)


def regions_ms(tag):
    acc = np.zeros(capi.N_TIMED_STAGES)
    for _ in range(2):
        obj.step(capi.STAGE_REGIONS)
    for _ in range(10):
        acc += obj.step(capi.STAGE_REGIONS)["stage_ms"]
    print(tag, {capi.STAGE_NAMES[i]: round(float(acc[i] / 10), 4) for i in range(capi.N_TIMED_STAGES) if acc[i] > 0})


def census(tag):
    _, _, flg, _, info = obj.download(sdf=False, types=False, labels=False)
    multi = np.nonzero(info["region_count"] > 1)[0]
    print(tag, "chunks with several regions:", len(multi), "region counts:", np.bincount(info["region_count"][multi]).nonzero()[0].tolist()[:20])
    stats = []
    f = np.asarray(flg).reshape(-1, 16, 16, 16)
    for c in multi:
        ne = (f[c] & 1) == 0
        lower = np.zeros_like(ne)
        lower[1:] |= ne[:-1]; lower[:, 1:] |= ne[:, :-1]; lower[:, :, 1:] |= ne[:, :, :-1]
        src = ne & ~lower
        stats.append((int(c), int(ne.sum()), int(src.sum()), int(info["region_count"][c])))
    stats.sort(key=lambda s: -s[2])
    print("  (chunk, non-empty voxels, sources, regions) most sources first:", stats[:10])
    print("  total sources over these chunks:", sum(s[2] for s in stats))


regions_ms("before the edit")
census("before the edit:")
c = (np.array([0.5 * (a + b) for a, b in obj.update_occupied_voxel_ranges()], dtype=np.float32) + np.array([110.0, 6.0, -4.0], dtype=np.float32) * np.float32(scale))
r = obj.absorb_sphere(c, 15.0 * scale + 2.0, 15.0 * scale, want_invalidated=True)
print("edit:", {k: (v if np.isscalar(v) else None) for k, v in r.items() if k in ("touched_chunks", "removed_chunks")})
census("after the edit:")
regions_ms("after the edit")

import ctypes as C
lib = capi.lib()
if hasattr(lib, "ivx_debug_exact_trace"):
    buf = np.zeros((64, 8), dtype=np.uint64)
    assert lib.ivx_debug_exact_trace(buf.ctypes.data_as(C.c_void_p)) == 0
    for row in buf:
        if row[0]:
            d = (row[1:6].astype(np.int64) - row[:5].astype(np.int64)) * 0.01
            print("exact numbering of one chunk (us): pointers+jumping %.2f, event listing %.2f, events applied %.2f, flatten+ids %.2f, keys+numbers+labels %.2f | events %d" % (*d, int(row[7])))
