"""Many objects per call (`ivx_voxel_step_many`, `ivx_absorb_sphere_many`, `ivx_mesh_sync_many`; csrc/many.hpp): the fragments of one body —
Voronoi cells of a lattice of fracture points, copied out with `ivx_copy_polyhedra` — stepped, edited and re-meshed TOGETHER, every object
against the oracle's result for that object on its own (the reference loops over its objects: impact_voxel/src/lib.rs:729-733) and against
the single-object calls."""
import numpy as np
import pytest

import oracle_lib as ol
import parity_util as pu
from impact_amd import capi, many, scenes
from impact_amd import fracturing as fr
from impact_amd.voxel import VoxelObjectMesh
from test_gpu_mesh_sync import assert_synced_meshes_equal
from test_gpu_split import assert_objects_equal, build

pytestmark = pytest.mark.gpu


def lattice_fragments(ctx, graph, n_per_axis, jitter_seed=7):
    """the body's grid box cut into the Voronoi cells of a jittered lattice: (oracle object, gpu object) per non-empty cell"""
    o, g = build(ctx, graph)
    cc = np.asarray(o.chunk_counts, dtype=np.float32) * 16.0
    rng = np.random.default_rng(jitter_seed)
    ax = [(np.arange(n_per_axis) + 0.5) * (c / n_per_axis) for c in cc]
    pts = np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3) + rng.uniform(-3.0, 3.0, (n_per_axis ** 3, 3))
    sets, tets = fr.fragment_plane_sets(pts.astype(np.float32), np.array([0, 0, 0, cc[0], cc[1], cc[2]], dtype=np.float32))
    res = g.copy_polyhedra([s[2] for s in sets], [s[1] for s in sets])
    pairs = []
    for (v, planes, bb), (rc, child, off) in zip(sets, res):
        rco, co, org = o.clip_polyhedron(planes, bb, copy=True)
        assert rc == rco
        if rc == 1:
            assert tuple(off) == tuple(org)
            pairs.append((co, child))
    tets.close()
    g.close()
    return pairs


def test_fragments_stepped_edited_and_synced_together(ctx):
    pairs = lattice_fragments(ctx, scenes.sphere_scene(44.0), 3)
    assert len(pairs) >= 20
    os_, gs = [p[0] for p in pairs], [p[1] for p in pairs]
    dens = np.linspace(0.5, 2.0, 256).astype(np.float32)
    for g in gs:
        g.set_densities(dens)
    # ---- one step for all: derived state, regions, occupied ranges, mesh, moments
    res = many.voxel_step_many(gs, capi.STAGE_ALL & ~capi.STAGE_SAMPLE)
    oms, gms = [], []
    for k, (o, g) in enumerate(pairs):
        assert_objects_equal(o, g, f"fragment {k} after the step of all: ")
        om = o.mesh()
        assert (int(res[k]["mesh"]["n_vertices"]), int(res[k]["mesh"]["n_indices"])) == (om.positions.shape[0], om.indices.shape[0])
        _, o64 = o.inertia(dens)
        g64 = np.asarray(res[k]["moments"]["m64"], dtype=np.float64)
        assert np.all(np.abs(g64 - o64) <= 1e-5 * np.maximum(np.abs(o64), 1e-300) + 1e-12), k
        n_regions, _ = o.region_labels()
        assert int(res[k]["region_count"]) == n_regions
        gm = VoxelObjectMesh(g)
        gm.counts = res[k]["mesh"].copy()
        pos, nrm, idx, im, sub = gm.download()
        np.testing.assert_array_equal(idx, om.indices)
        np.testing.assert_array_equal(pos.view(np.uint32), om.positions.view(np.uint32))
        np.testing.assert_array_equal(nrm.view(np.uint32), om.normals.view(np.uint32))
        np.testing.assert_array_equal(im, om.index_materials)
        oms.append(ol.OracleMeshHandle(o))
        gms.append(gm)
    # ---- frames: one absorbing sphere per object, the incremental remesh of what it invalidated, the moments — all objects per call
    for o in os_:
        o.update_occupied_voxel_ranges()
    for frame in range(3):
        centers, radii = [], []
        for o in os_:
            occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float32)
            c = 0.5 * (occ[:, 0] + occ[:, 1])
            c[frame % 3] = occ[frame % 3, 1] - 1.0  # at the object's upper face along the frame's axis
            centers.append(c)
            radii.append(3.0 + frame)
        ros = [o.absorb_sphere(c, r + 2.0, r, dens) for o, c, r in zip(os_, centers, radii)]
        rgs = many.absorb_sphere_many(gs, centers, [r + 2.0 for r in radii], radii, dens)
        for k, (ro, rg) in enumerate(zip(ros, rgs)):
            np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"], err_msg=f"fragment {k}, frame {frame}")
            assert (rg["touched_chunks"], rg["removed_chunks"], rg["emptied_voxels"]) == (ro["touched_chunks"], ro["removed_chunks"], int(ro["emptied_by_type"].sum()))
            scale = np.maximum(np.abs(ro["removed64"]), 1e-300)
            assert np.all(np.abs(rg["removed_moments"] - ro["removed64"]) <= 1e-5 * scale + 1e-9), (k, frame, rg["emptied_voxels"], rg["removed_moments"][0], ro["removed64"][0])
            pu.assert_edited_objects_equal(os_[k], gs[k], f"fragment {k}, frame {frame}: ", with_mesh=False)
        for om, ro in zip(oms, ros):
            om.sync(ro["invalidated"])
        many.mesh_sync_many(gms, [rg["invalidated"] for rg in rgs])
        for k in range(len(gs)):
            assert_synced_meshes_equal(oms[k].get(), gms[k].download())
        mom = many.voxel_step_many(gs, capi.STAGE_INERTIA)
        for k, o in enumerate(os_):
            _, o64 = o.inertia(dens)
            g64 = np.asarray(mom[k]["moments"]["m64"], dtype=np.float64)
            assert np.all(np.abs(g64 - o64) <= 1e-5 * np.maximum(np.abs(o64), 1e-300) + 1e-12), (k, frame)
    for g in gs:
        g.close()


def test_many_calls_equal_the_single_object_calls(ctx):
    """the same objects twice: one set through the `_many` calls, the other object by object — every buffer equal"""
    a = lattice_fragments(ctx, scenes.asteroid_scene(0.4), 2, jitter_seed=3)
    b = lattice_fragments(ctx, scenes.asteroid_scene(0.4), 2, jitter_seed=3)
    ga, gb = [p[1] for p in a], [p[1] for p in b]
    assert len(ga) == len(gb) >= 4
    stages = capi.STAGE_ALL & ~capi.STAGE_SAMPLE & ~capi.STAGE_INERTIA
    ra = many.voxel_step_many(ga, stages)
    rb = [g.step(stages) for g in gb]
    for k in range(len(ga)):
        for f in ("region_count",):
            assert int(ra[k][f]) == int(rb[k][f])
        np.testing.assert_array_equal(np.asarray(ra[k]["occupied"]), np.asarray(rb[k]["occupied"]))
        for x, y in zip(ga[k].download(), gb[k].download()):
            np.testing.assert_array_equal(x, y)
        ma, mb = VoxelObjectMesh(ga[k]), VoxelObjectMesh(gb[k])
        ma.counts, mb.counts = ra[k]["mesh"].copy(), rb[k]["mesh"].copy()
        for x, y in zip(ma.download(), mb.download()):
            np.testing.assert_array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8))
    for g in ga + gb:
        g.close()


def test_many_calls_at_the_edges(ctx):
    """no object at all, one object, an object listed twice, spheres that miss their objects next to spheres that hit, and a step with the
    sample stage (which runs object by object inside the call)"""
    lib = capi.lib()
    # nothing to do is not an error (a manager without voxel objects)
    assert many.voxel_step_many([], capi.STAGE_ALL).size == 0
    assert many.absorb_sphere_many([], np.zeros((0, 3)), [], []) == []
    assert many.mesh_sync_many([], []) == []
    pairs = lattice_fragments(ctx, scenes.sphere_scene(30.0), 2, jitter_seed=5)
    os_, gs = [p[0] for p in pairs], [p[1] for p in pairs]
    assert len(gs) >= 4
    stages = capi.STAGE_ALL & ~capi.STAGE_SAMPLE
    # one object: the same as the single-object call
    r1 = many.voxel_step_many(gs[:1], stages)
    assert_objects_equal(os_[0], gs[0], "one object: ")
    assert int(r1[0]["region_count"]) == os_[0].region_labels()[0]
    # an object listed twice is refused before anything is enqueued
    with pytest.raises(capi.IvxError):
        many.voxel_step_many([gs[0], gs[1], gs[0]], stages)
    res = many.voxel_step_many(gs, stages)
    gms = []
    for g, r in zip(gs, res):
        m = VoxelObjectMesh(g)
        m.counts = r["mesh"].copy()
        gms.append(m)
    oms = [ol.OracleMeshHandle(o) for o in os_]
    for o in os_:
        o.update_occupied_voxel_ranges()
    # every other sphere lies far outside its object: those objects report an edit that touched nothing, the others are edited
    centers, radii = [], []
    for k, o in enumerate(os_):
        occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float32)
        c = 0.5 * (occ[:, 0] + occ[:, 1])
        if k % 2:
            c = c + 500.0
        else:
            c[0] = occ[0, 1] - 1.0
        centers.append(c)
        radii.append(4.0)
    ros = [o.absorb_sphere(c, r + 2.0, r) for o, c, r in zip(os_, centers, radii)]
    rgs = many.absorb_sphere_many(gs, centers, [r + 2.0 for r in radii], radii)
    for k, (ro, rg) in enumerate(zip(ros, rgs)):
        np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"], err_msg=f"object {k}")
        assert (rg["touched_chunks"], rg["emptied_voxels"]) == (ro["touched_chunks"], int(ro["emptied_by_type"].sum()))
        if k % 2:
            assert rg["touched_chunks"] == 0 and not rg["invalidated"].any()
        pu.assert_edited_objects_equal(os_[k], gs[k], f"object {k}: ", with_mesh=False)
    for om, ro in zip(oms, ros):
        om.sync(ro["invalidated"])
    many.mesh_sync_many(gms, [rg["invalidated"] for rg in rgs])
    for k in range(len(gs)):
        assert_synced_meshes_equal(oms[k].get(), gms[k].download())
    # one capsule per object through the batched call
    segs, vecs = [], []
    for o in os_:
        occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float32)
        a = 0.5 * (occ[:, 0] + occ[:, 1])
        a[2] = occ[2, 1] - 1.0
        segs.append(a)
        vecs.append(np.array([3.0, -2.0, 0.5], dtype=np.float32))
    ros = [o.absorb_capsule(a, v, 4.5, 2.5) for o, a, v in zip(os_, segs, vecs)]
    rgs = many.absorb_capsule_many(gs, segs, vecs, [4.5] * len(gs), [2.5] * len(gs))
    for k, (ro, rg) in enumerate(zip(ros, rgs)):
        np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"], err_msg=f"capsule, object {k}")
        assert (rg["touched_chunks"], rg["emptied_voxels"]) == (ro["touched_chunks"], int(ro["emptied_by_type"].sum()))
        pu.assert_edited_objects_equal(os_[k], gs[k], f"capsule, object {k}: ", with_mesh=False)
    for om, ro in zip(oms, ros):
        om.sync(ro["invalidated"])
    many.mesh_sync_many(gms, [rg["invalidated"] for rg in rgs])
    for k in range(len(gs)):
        assert_synced_meshes_equal(oms[k].get(), gms[k].download())
    # the recorder by hand around the single-object halves: the same results as the blocking calls
    ctr = []
    for o in os_:
        occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float32)
        c = 0.5 * (occ[:, 0] + occ[:, 1])
        c[1] = occ[1, 1] - 1.0
        ctr.append(c)
    ros = [o.absorb_sphere(c, 5.0, 3.0) for o, c in zip(os_, ctr)]
    capi.check(lib.ivx_many_begin(ctx.h))
    for g, c in zip(gs, ctr):
        g.absorb_sphere_enqueue(c, 5.0, 3.0)
    capi.check(lib.ivx_many_flush(ctx.h))
    for k, (g, ro) in enumerate(zip(gs, ros)):
        rg = g.absorb_collect()
        np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"], err_msg=f"object {k}")
        pu.assert_edited_objects_equal(os_[k], g, f"recorded by hand, object {k}: ", with_mesh=False)
    for g in gs:
        g.close()


def test_a_call_on_another_context_is_not_recorded(ctx):
    """between ivx_many_begin(ctx A) and its flush, a call on an object of context B goes out on B's own stream (what A has recorded is
    flushed first, recording resumes afterwards): results of both as if nothing had been recorded"""
    from impact_amd.voxel import Context

    lib = capi.lib()
    other = Context(0)
    try:
        from test_gpu_mesh_sync import both

        oa, ga = both(ctx, scenes.sphere_scene(22.0))
        ob, gb = both(other, scenes.sphere_scene(18.0))

        def top(o):
            occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float32)
            c = 0.5 * (occ[:, 0] + occ[:, 1])
            c[0] = occ[0, 1] - 1.0
            return c

        ca, cb = top(oa), top(ob)
        ra, rb = oa.absorb_sphere(ca, 6.0, 4.0), ob.absorb_sphere(cb, 5.0, 3.0)
        capi.check(lib.ivx_many_begin(ctx.h))
        ga.absorb_sphere_enqueue(ca, 6.0, 4.0)   # recorded for context A
        gb.absorb_sphere_enqueue(cb, 5.0, 3.0)   # context B: flushes A's, runs unrecorded
        capi.check(lib.ivx_many_flush(ctx.h))
        xa, xb = ga.absorb_collect(), gb.absorb_collect()
        np.testing.assert_array_equal(xa["invalidated"], ra["invalidated"])
        np.testing.assert_array_equal(xb["invalidated"], rb["invalidated"])
        pu.assert_edited_objects_equal(oa, ga, "context A: ", with_mesh=False)
        pu.assert_edited_objects_equal(ob, gb, "context B: ", with_mesh=False)
        ga.close()
        gb.close()
    finally:
        other.close()


def test_step_many_with_the_sample_stage(ctx):
    """`ivx_voxel_step_many` with IVX_STAGE_SAMPLE: the sample stage has no twin and runs object by object inside the call; the rest merges"""
    from impact_amd.voxel import SDFVoxelGenerator, VoxelObject

    graphs = [scenes.sphere_scene(20.0), scenes.asteroid_scene(0.3), scenes.box_scene((30.0, 30.0, 30.0))]
    objs, refs = [], []
    for gr in graphs:
        gen = SDFVoxelGenerator(1.0, gr, 0)
        o = VoxelObject(ctx, gen.chunk_counts(), 1.0)
        o.set_sdf_program(gen)
        o.set_densities(np.ones(256, dtype=np.float32))
        objs.append(o)
        refs.append(pu.oracle_from_graph(gr, 1.0))
    res = many.voxel_step_many(objs, capi.STAGE_ALL)
    for k, (o, g) in enumerate(zip(refs, objs)):
        o.update_occupied_voxel_ranges()
        o.compute_all_derived_state()
        parity = pu.step_parity(o, g, res[k])
        assert parity["equal"], (k, parity)
        g.close()
