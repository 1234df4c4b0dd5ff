#!/usr/bin/env python3
"""Developer tool: where a many-object frame's time goes (bench.py's `fragments` leg, phase by phase, looped and merged)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impact_amd import capi, many, scenes
from impact_amd import fracturing as fr
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject, VoxelObjectMesh


def main():
    n_axis = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ctx = Context(0)
    gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(1.0), 0)
    body = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    body.set_sdf_program(gen)
    dens = np.ones(256, dtype=np.float32)
    body.set_densities(dens)
    body.step(capi.STAGE_ALL)
    cc = np.asarray(body.chunk_counts, dtype=np.float32) * 16.0
    rng = np.random.default_rng(11)
    ax = [(np.arange(n_axis) + 0.5) * (c / n_axis) for c in cc]
    pts = (np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3) + rng.uniform(-4.0, 4.0, (n_axis ** 3, 3))).astype(np.float32)
    sets, tets = fr.fragment_plane_sets(pts, np.array([0, 0, 0, cc[0], cc[1], cc[2]], dtype=np.float32))
    res = body.copy_polyhedra([s[2] for s in sets], [s[1] for s in sets])
    objs = [c for rc, c, _ in res if rc == 1]
    for o in objs:
        o.set_densities(dens)
    n = len(objs)
    stages0 = capi.STAGE_ALL & ~capi.STAGE_SAMPLE
    for rep in range(3):
        ctx.synchronize(); t0 = time.perf_counter()
        r = many.voxel_step_many(objs, stages0)
        ctx.synchronize(); t1 = time.perf_counter()
        print(f"step_many(all) rep {rep}: {1e3 * (t1 - t0):.3f} ms for {n} objects")
    meshes = []
    for o, rr in zip(objs, r):
        m = VoxelObjectMesh(o); m.counts = rr["mesh"].copy(); meshes.append(m)
    occ = [np.asarray(rr["occupied"], dtype=np.float32).reshape(-1)[6:].reshape(3, 2) for rr in r]
    many.mesh_sync_many(meshes, [np.zeros(o.n_chunks, dtype=np.uint8) for o in objs])
    for f in range(5):
        cs, rs = [], []
        for oc in occ:
            c = 0.5 * (oc[:, 0] + oc[:, 1]); c[f % 3] = oc[f % 3, 1] - 1.0 - 2.0 * (f // 3)
            cs.append(c.astype(np.float32)); rs.append(4.0 + (f % 3))
        ctx.synchronize(); t0 = time.perf_counter()
        e = many.absorb_sphere_many(objs, cs, [x + 2.0 for x in rs], rs, dens)
        t1 = time.perf_counter()
        many.mesh_sync_many(meshes, [x["invalidated"] for x in e])
        t2 = time.perf_counter()
        many.voxel_step_many(objs, capi.STAGE_INERTIA)
        t3 = time.perf_counter()
        print(f"frame {f}: absorb_many {1e3 * (t1 - t0):.3f}  sync_many {1e3 * (t2 - t1):.3f}  inertia_many {1e3 * (t3 - t2):.3f} ms")


if __name__ == "__main__":
    main()
