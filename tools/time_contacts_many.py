#!/usr/bin/env python3
"""Developer tool: ivx_voxel_object_contacts_many / ivx_mutual_voxel_object_contacts_many on the fragments of bench.py's `fragments_frame`,
call by call (IVX_MANY_TRACE=1 in the environment prints the host time of each recorded phase). usage: time_contacts_many.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from impact_amd import capi, many, scenes
from impact_amd import fracturing as fr
from impact_amd.voxel import Context, VoxelObjectMesh
import bench

ctx = Context(0)
graph = scenes.asteroid_scene(1.0)
dens = np.ones(256, dtype=np.float32)
_, body = bench.make_object(ctx, graph)
body.step(capi.STAGE_ALL)
cc = np.asarray(body.chunk_counts, dtype=np.float32) * 16.0
rng = np.random.default_rng(11)
n_axis = 5
ax = [(np.arange(n_axis) + 0.5) * (c / n_axis) for c in cc]
pts = (np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3) + rng.uniform(-4.0, 4.0, (n_axis ** 3, 3))).astype(np.float32)
sets, tets = fr.fragment_plane_sets(pts, np.array([0, 0, 0, cc[0], cc[1], cc[2]], dtype=np.float32))
res = body.copy_polyhedra([s_[2] for s_ in sets], [s_[1] for s_ in sets])
objs = [child for rc, child, _ in res if rc == 1]
for o_ in objs:
    o_.set_densities(dens)
n = len(objs)
rs = many.voxel_step_many(objs, capi.STAGE_ALL & ~capi.STAGE_SAMPLE)
occ = [np.asarray(r_["occupied"], dtype=np.float32).reshape(-1)[6:].reshape(3, 2) for r_ in rs]
up = np.array([0.0, 1.0, 0.0], dtype=np.float32)
q = many.collidable_queries(n)
for k in range(n):
    q[k]["mode"], q[k]["shape3"], q[k]["shape1"], q[k]["response"] = 1, up, float(occ[k][1, 0] + 3.0), (0.2, 0.7, 0.5)
    q[k]["collidable_id_a"], q[k]["collidable_id_b"], q[k]["body_a"], q[k]["body_b"] = 1000 + k, 7, k, 0x80000000
for rep in range(6):
    ctx.synchronize()
    t0 = time.perf_counter()
    cb, off = many.voxel_object_contacts_many(objs, q)
    t1 = time.perf_counter()
    print(f"contacts_many: {1e3 * (t1 - t0):.4f} ms, {len(cb)} contacts", file=sys.stderr)
for o_ in objs:
    o_.collision_probes_recompute()
ident = np.array([0.0, 0.0, 0.0, 1.0], dtype=np.float32)
zero3 = np.zeros(3, dtype=np.float32)
pairs = []
for k in range(n - 1):
    ca, cb_ = 0.5 * (occ[k][:, 0] + occ[k][:, 1]), 0.5 * (occ[k + 1][:, 0] + occ[k + 1][:, 1])
    tb = np.array([occ[k + 1][0, 0] - occ[k][0, 1] + 6.0, cb_[1] - ca[1], cb_[2] - ca[2]], dtype=np.float32)
    pairs.append(dict(a=objs[k], b=objs[k + 1], rotation_a=ident, translation_a=zero3, center_of_mass_a=ca, rotation_b=ident, translation_b=tb, center_of_mass_b=cb_,
                      collidable_id_a=500 + k, collidable_id_b=501 + k, body_a=k, body_b=k + 1, response=(0.2, 0.7, 0.5)))
qb = many.mutual_queries(pairs)
for rep in range(6):
    ctx.synchronize()
    t0 = time.perf_counter()
    got, off = many.mutual_voxel_object_contacts_many(qb)
    t1 = time.perf_counter()
    print(f"mutual_many: {1e3 * (t1 - t0):.4f} ms, {len(got)} contacts", file=sys.stderr)
