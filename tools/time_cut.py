#!/usr/bin/env python3
"""The fracture event, timed piece by piece (developer tool; bench.py's `fragments` leg has the headline figures): the config-2 body (256^3)
cut into the Voronoi cells of a jittered 5^3 lattice — `ivx_copy_polyhedra` alone (the C call), the Python wrappers of the children, their
density tables, the first step of all fragments (`ivx_voxel_step_many`: first meshes, buffers from one allocation) — and the oracle doing the
cut with the fragments in parallel on the host cores, as fracturing.rs:1047-1189 does. usage: time_cut.py [reps] [--oracle]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from impact_amd import capi, many, scenes  # noqa: E402
from impact_amd import fracturing as fr  # noqa: E402
from impact_amd.capi import ptr  # noqa: E402
from impact_amd.voxel import Context  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
ctx = Context(0)
graph = scenes.asteroid_scene(1.0)
_, body = bench.make_object(ctx, graph)
body.step(capi.STAGE_ALL)
cc = np.asarray(body.chunk_counts, dtype=np.float32) * 16.0
rng = np.random.default_rng(11)
ax = [(np.arange(5) + 0.5) * (c / 5) for c in cc]
pts = (np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3) + rng.uniform(-4.0, 4.0, (125, 3))).astype(np.float32)
sets, tets = fr.fragment_plane_sets(pts, np.array([0, 0, 0, cc[0], cc[1], cc[2]], dtype=np.float32))
plane_sets = [np.ascontiguousarray(s[1], dtype=np.float32).reshape(-1, 4) for s in sets]
planes = np.ascontiguousarray(np.concatenate(plane_sets))
counts = np.array([len(p) for p in plane_sets], dtype=np.uint32)
bbs = np.ascontiguousarray(np.asarray([s[2] for s in sets], dtype=np.float32).reshape(-1, 6))
n = len(sets)
lib = capi.lib()
dens = np.ones(256, dtype=np.float32)
for rep in range(reps):
    children = (C.c_void_p * n)()
    origins = np.zeros((n, 3), dtype=np.uint32)
    outcomes = np.zeros(n, dtype=np.int32)
    ctx.synchronize()
    t0 = time.perf_counter()
    capi.check(lib.ivx_copy_polyhedra(body.h, ptr(planes), ptr(counts), ptr(bbs), n, children, ptr(origins), ptr(outcomes)))
    t1 = time.perf_counter()
    objs = [body._wrap_child(C.c_void_p(children[f])) for f in range(n) if outcomes[f] == 1]
    t2 = time.perf_counter()
    for o in objs:
        o.set_densities(dens)
    ctx.synchronize()
    t3 = time.perf_counter()
    res = many.voxel_step_many(objs, capi.STAGE_ALL & ~capi.STAGE_SAMPLE)
    ctx.synchronize()
    t4 = time.perf_counter()
    res = many.voxel_step_many(objs, capi.STAGE_ALL & ~capi.STAGE_SAMPLE)
    ctx.synchronize()
    t5 = time.perf_counter()
    print(f"rep {rep}: {len(objs)} fragments of {n} cells | ivx_copy_polyhedra {1e3 * (t1 - t0):.3f} ms | wrappers {1e3 * (t2 - t1):.3f} | density tables {1e3 * (t3 - t2):.3f} | "
          f"first step of all {1e3 * (t4 - t3):.3f} | second step of all {1e3 * (t5 - t4):.3f} | triangles {int(res['mesh']['n_indices'].sum()) // 3}")
    t6 = time.perf_counter()
    for o in objs:
        o.close()
    print(f"        closing them {1e3 * (time.perf_counter() - t6):.3f} ms")
if "--oracle" in sys.argv:
    import oracle_lib as ol
    from concurrent.futures import ThreadPoolExecutor

    o = ol.OracleObject.from_sdf(graph, 1.0, 0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    workers = min(16, len(os.sched_getaffinity(0)))

    def one(k):
        rc, co, _ = o.clip_polyhedron(sets[k][1], sets[k][2], copy=True)
        return rc

    t0 = time.perf_counter()
    with ThreadPoolExecutor(workers) as ex:
        rcs = list(ex.map(one, range(n)))
    t1 = time.perf_counter()
    print(f"oracle: the same cut, fragments in parallel on {workers} host threads (ctypes releases the GIL): {1e3 * (t1 - t0):.1f} ms, {sum(1 for r in rcs if r == 1)} fragments")
tets.close()
body.close()
ctx.close()
