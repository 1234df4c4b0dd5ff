#!/usr/bin/env python3
"""Developer tool: the bench's edit followed by the incremental remesh (ivx_mesh_sync), enqueue and collect timed apart."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impact_amd import capi, scenes
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject, VoxelObjectMesh

scale = 2.05
ctx = Context(0)
gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(scale), 0)
obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
obj.set_sdf_program(gen)
dens = np.ones(256, dtype=np.float32)
obj.set_densities(dens)
t_e, t_q, t_c, t_ee, t_ec = [], [], [], [], []
for rep in range(8):
    r = obj.step(capi.STAGE_ALL)
    mesh = VoxelObjectMesh(obj); mesh.counts = r["mesh"].copy()
    mesh.sync_with_voxel_object(np.zeros(obj.n_chunks, dtype=np.uint8))
    c = (np.array([0.5 * (a + b) for a, b in obj.update_occupied_voxel_ranges()], dtype=np.float32) + np.array([110.0, 6.0, -4.0], dtype=np.float32) * np.float32(scale))
    ctx.synchronize()
    t0 = time.perf_counter()
    obj.absorb_sphere_enqueue(c, 15.0 * scale + 2.0, 15.0 * scale, dens)
    t1 = time.perf_counter()
    e = obj.absorb_collect()
    t2 = time.perf_counter()
    mesh.sync_enqueue(e["invalidated"])
    t3 = time.perf_counter()
    mesh.sync_collect()
    t4 = time.perf_counter()
    t_ee.append(t1 - t0); t_ec.append(t2 - t1); t_q.append(t3 - t2); t_c.append(t4 - t3)
f = lambda x: round(1e3 * float(np.mean(x[2:])), 4)
print("edit enqueue ms", f(t_ee), " edit collect ms", f(t_ec), " sync enqueue ms", f(t_q), " sync collect ms", f(t_c), " invalidated", int(np.count_nonzero(e["invalidated"])))

import ctypes as C
lib = capi.lib(); lib.ivx_grid_device_ptr.restype = C.c_void_p
pp = lib.ivx_grid_device_ptr(obj.h, 9)
hip = C.CDLL("libamdhip64.so")
w = np.zeros(4, dtype=np.uint32)
assert hip.hipMemcpy(w.ctypes.data_as(C.c_void_p), C.c_void_p(pp), 16, 2) == 0
print("chunks handed to the general pass by the last sync:", int(w[0]))
sub = mesh.download()[4]
print("submeshes:", len(sub), "largest vertex counts:", np.sort(np.asarray([int(s[14]) for s in sub]))[-5:].tolist() if len(sub) and not hasattr(sub, "dtype") or sub.dtype.names is None else np.sort(sub["vertex_count"])[-5:].tolist())
sdf, typ, flg, lab, info = obj.download()
typ = np.asarray(typ).reshape(-1, 4096); flg = np.asarray(flg).reshape(-1, 4096); sdf = np.asarray(sdf).reshape(-1, 4096)
inv = np.nonzero(e["invalidated"])[0]
odd = []
for c in inv:
    ne = (flg[c] & 1) == 0
    neg = sdf[c].view(np.int8) < 0
    t_ne = np.unique(typ[c][ne]) if ne.any() else []
    t_neg = np.unique(typ[c][neg]) if neg.any() else []
    if len(t_ne) > 1 or len(t_neg) > 1 or (ne != neg).any():
        odd.append((int(c), int(info["kind"][c]), [int(x) for x in t_ne], [int(x) for x in t_neg], int((ne != neg).sum())))
print("invalidated chunks whose non-empty / negative voxels hold several types, or where non-empty != negative:", odd[:8], len(odd))
pl = lib.ivx_grid_device_ptr(obj.h, 10)
hl = np.zeros(4, dtype=np.uint32)
assert hip.hipMemcpy(hl.ctypes.data_as(C.c_void_p), C.c_void_p(pl), 16, 2) == 0
have = set()
for srow in sub:
    ci = srow["chunk_indices"] if sub.dtype.names else srow[:3]
    have.add((int(ci[0]) * obj.chunk_counts[1] + int(ci[1])) * obj.chunk_counts[2] + int(ci[2]))
recs = [int(c) for c in inv if int(c) in have]
c = recs[int(hl[0])]
cc = obj.chunk_counts
ci, cj, ck = c // (cc[1] * cc[2]), (c // cc[2]) % cc[1], c % cc[2]
print("first entry of the hard list:", int(hl[0]), "-> chunk", c, (ci, cj, ck), "kind", int(info["kind"][c]), "flags", hex(int(info["flags"][c])))
for d, (a, b, g_) in enumerate([(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]):
    n = ((ci + a) * cc[1] + cj + b) * cc[2] + ck + g_
    ne = (flg[n] & 1) == 0
    print("  neighbour", (a, b, g_), "chunk", n, "kind", int(info["kind"][n]), "uniform_type", int(info["uniform_type"][n]), "types of non-empty voxels", np.unique(typ[n][ne]).tolist()[:4], "non-empty", int(ne.sum()))
pos, nrm, idx, im, sub2 = mesh.download()
for srow in sub2:
    cix = srow["chunk_indices"]
    if (int(cix[0]), int(cix[1]), int(cix[2])) == (ci, cj, ck):
        io, ic = int(srow["index_offset"]), int(srow["index_count"])
        u, cnt = np.unique(im[io:io + ic], axis=0, return_counts=True)
        print("index materials of that chunk's submesh (8 bytes each: 4 material indices, 4 weights):", u.tolist()[:6], cnt.tolist()[:6])
