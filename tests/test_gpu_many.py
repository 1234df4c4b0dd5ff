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


def _stats(ctx):
    out = np.zeros(3, dtype=np.uint64)
    capi.check(capi.lib().ivx_many_stats(ctx.h, capi.ptr(out)))
    return [int(x) for x in out]


def test_a_batch_recorded_by_hand_merges(ctx):
    """the bare bracket (ivx_many_begin ... ivx_many_flush around `_enqueue` calls) keeps a chain per object and merges them front by front:
    far fewer launches issued than recorded, results as the blocking calls give them"""
    from test_gpu_mesh_sync import both

    lib = capi.lib()
    pairs = [both(ctx, scenes.sphere_scene(16.0 + 2.0 * k)) for k in range(6)]
    ctr = []
    for o, _ in pairs:
        occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float32)
        c = 0.5 * (occ[:, 0] + occ[:, 1])
        c[2] = occ[2, 1] - 1.0
        ctr.append(c)
    for (o, g), c in zip(pairs, ctr):  # (an object's first edit allocates its edit buffers, with waits on the stream: not what is measured here;
        # nor is the upload of a density table other than the resident one, a copy in the middle of the chain)
        g.set_densities(np.ones(256, dtype=np.float32))
        o.absorb_sphere(c + np.float32(1.5), 4.0, 2.0)
        g.absorb_sphere(c + np.float32(1.5), 4.0, 2.0)
    ros = [o.absorb_sphere(c, 5.0, 3.0) for (o, _), c in zip(pairs, ctr)]
    rec0, iss0, _ = _stats(ctx)
    capi.check(lib.ivx_many_begin(ctx.h))
    for (_, g), c in zip(pairs, ctr):
        g.absorb_sphere_enqueue(c, 5.0, 3.0)
    capi.check(lib.ivx_many_flush(ctx.h))
    rec1, iss1, _ = _stats(ctx)
    recorded, issued = rec1 - rec0, iss1 - iss0
    assert recorded >= 6 * 5, (recorded, issued)  # (every object records its chain of five or more launches)
    assert issued * 3 <= recorded, f"{recorded} launches recorded, {issued} issued: the objects' chains did not merge"
    for k, ((o, g), ro) in enumerate(zip(pairs, ros)):
        rg = g.absorb_collect()
        np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"], err_msg=f"object {k}")
        pu.assert_edited_objects_equal(o, g, f"recorded by hand, object {k}: ", with_mesh=False)
        g.close()


def test_calls_without_a_guard_on_another_context_are_not_recorded(ctx):
    """ivx_inertia / ivx_derive_state / ivx_label_regions of an object of context B between ivx_many_begin(A) and its flush: their twinned launches
    (moment sweep, chunk pre-pass, derive sweep) must go out on B's stream, not into A's batch — the check sits at the capture itself"""
    from impact_amd.voxel import Context, VoxelObjectInertialPropertyManager
    from test_gpu_mesh_sync import both

    lib = capi.lib()
    other = Context(0)
    try:
        oa, ga = both(ctx, scenes.sphere_scene(20.0))
        ob, gb = both(other, scenes.asteroid_scene(0.25))
        dens = np.linspace(0.5, 2.0, 256).astype(np.float32)
        occ = np.array(oa.info()["occupied_voxel_ranges"], dtype=np.float32)
        ca = 0.5 * (occ[:, 0] + occ[:, 1])
        ca[0] = occ[0, 1] - 1.0
        ra = oa.absorb_sphere(ca, 6.0, 4.0)
        _, want = ob.inertia(dens)
        rec0 = _stats(ctx)[0]
        capi.check(lib.ivx_many_begin(ctx.h))
        ga.absorb_sphere_enqueue(ca, 6.0, 4.0)  # recorded for A
        gb.compute_all_derived_state()          # B: chunk pre-pass + derive sweep + regions, on B's stream
        got = VoxelObjectInertialPropertyManager.initialized_from(gb, dens).m64  # B: moment sweep
        capi.check(lib.ivx_many_flush(ctx.h))
        assert np.all(np.abs(got - want) <= 1e-5 * np.maximum(np.abs(want), 1e-300) + 1e-12)
        xa = ga.absorb_collect()
        np.testing.assert_array_equal(xa["invalidated"], ra["invalidated"])
        pu.assert_edited_objects_equal(oa, ga, "context A: ", with_mesh=False)
        assert_objects_equal(ob, gb, "context B: ")
        assert _stats(other)[0] == 0, "launches of context B's object were recorded"
        assert _stats(ctx)[0] > rec0
        ga.close()
        gb.close()
    finally:
        other.close()


def test_an_object_that_fails_in_the_middle_leaves_nobody_pending(ctx):
    """`ivx_mesh_sync_many` / `ivx_absorb_sphere_many` with an object the call refuses in the middle of the list: the call fails, and every
    object — the ones enqueued before the failure too — takes the next call as if nothing had happened"""
    from test_gpu_mesh_sync import both

    pairs = [both(ctx, scenes.sphere_scene(15.0 + 2.0 * k)) for k in range(5)]
    gs = [g for _, g in pairs]
    ctr = []
    for o, _ in pairs:
        occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float32)
        c = 0.5 * (occ[:, 0] + occ[:, 1])
        c[1] = occ[1, 1] - 1.0
        ctr.append(c)
    bad_radius = [5.0, 5.0, -1.0, 5.0, 5.0]  # (object 2: a radius the edit refuses)
    with pytest.raises(Exception):
        many.absorb_sphere_many(gs, ctr, bad_radius, [3.0] * 5)
    # objects 0 and 1 were enqueued before the failure: their edits ran (the oracle follows), their results were discarded — and nobody is
    # "in flight": the same edit, valid this time, goes through for all five
    for k in (0, 1):
        pairs[k][0].absorb_sphere(ctr[k], 5.0, 3.0)
    ros = [o.absorb_sphere(c, 5.0, 3.0) for (o, _), c in zip(pairs, ctr)]
    rs = many.absorb_sphere_many(gs, ctr, [5.0] * 5, [3.0] * 5)
    for k, ((o, g), ro, rg) in enumerate(zip(pairs, ros, rs)):
        np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"], err_msg=f"object {k}")
        pu.assert_edited_objects_equal(o, g, f"after the failed batch, object {k}: ", with_mesh=False)
    # a second failure kind: an object listed twice is refused before anything is enqueued
    with pytest.raises(Exception):
        many.mesh_sync_many([VoxelObjectMesh(gs[0]), VoxelObjectMesh(gs[0])], [rs[0]["invalidated"], rs[0]["invalidated"]])
    for g in gs:
        g.close()


def test_contexts_come_and_go_around_many_calls(ctx):
    """fifty contexts made, used for many-object calls and shut down: the recorder and its staging ring belong to the context and go with it
    (device and pinned memory flat), and a context made afterwards records on blocks of its own"""
    import ctypes

    from impact_amd.voxel import Context
    from test_gpu_mesh_sync import both

    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    def round_trip():
        c = Context(0)
        try:
            pairs = [both(c, scenes.sphere_scene(14.0 + k)) for k in range(3)]
            gs = [g for _, g in pairs]
            ctr = []
            for o, _ in pairs:
                occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float32)
                x = 0.5 * (occ[:, 0] + occ[:, 1])
                x[1] = occ[1, 1] - 1.0
                ctr.append(x)
            ros = [o.absorb_sphere(x, 4.0, 2.5) for (o, _), x in zip(pairs, ctr)]
            rs = many.absorb_sphere_many(gs, ctr, [4.0] * 3, [2.5] * 3)
            for ro, rg in zip(ros, rs):
                np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
            for g in gs:
                g.set_densities(np.ones(256, dtype=np.float32))
            many.voxel_step_many(gs, capi.STAGE_ALL & ~capi.STAGE_SAMPLE)
            for g in gs:
                g.close()
        finally:
            c.close()

    for _ in range(3):  # (first uses: code objects, the runtime's own pools)
        round_trip()
    before = free_bytes()
    for _ in range(50):
        round_trip()
    after = free_bytes()
    assert before - after < (8 << 20), f"device memory shrank by {before - after} bytes over 50 contexts"


@pytest.mark.parametrize("seed", pu.fuzz_seeds([13, 14, 20, 21, 26, 28]))
def test_random_batches_of_contacts_pairs_and_probe_syncs(ctx, seed):
    """the batched forms of a frame's other per-object loops on random data — a handful of bodies of random shape, extent and pose; a random
    collidable (sphere / plane / capsule) per body, a random list of pairs (bodies repeated, pairs apart, pairs deep inside one another), a
    random bite per body with its mesh and probe sync — against the single-object calls, byte for byte (those are held to the oracle in
    test_gpu_contacts.py, test_gpu_collide.py and test_gpu_mesh_sync.py)"""
    import test_gpu_collide as tc

    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 7))
    shapes = []
    for _ in range(n):
        kind = int(rng.integers(0, 3))
        if kind == 0:
            shapes.append((scenes.sphere_scene(float(rng.uniform(9.0, 26.0))), [1.0, 0.5][int(rng.integers(0, 2))]))
        elif kind == 1:
            shapes.append((scenes.box_scene(tuple(float(x) for x in rng.uniform(6.0, 34.0, 3))), 1.0))
        else:
            shapes.append((scenes.asteroid_scene(float(rng.uniform(0.2, 0.35))), 1.0))
    import test_gpu_contacts as tcon

    both_ = [tc.both(ctx, g_, e_) for g_, e_ in shapes]
    orcs, objs = [b[0] for b in both_], [b[1] for b in both_]
    o_probes = []  # (the oracle's probe lists: its mutual contacts take them as inputs)
    for o, g in both_:
        g.collision_probes_recompute()
        o_probes.append(o.collision_probes(o.mesh()))
    # poses: centres of mass on a line with random gaps (some overlap, some do not), random rotations
    pose, x = [], 0.0
    for g in objs:
        occ = np.asarray(g.update_occupied_voxel_ranges(), dtype=np.float64)
        com = (0.5 * (occ[:, 0] + occ[:, 1]) * g.voxel_extent).astype(np.float32)
        half = float(0.5 * (occ[:, 1] - occ[:, 0]).max() * g.voxel_extent)
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = rng.uniform(-1.5, 1.5)
        q = np.array([*(ax * np.sin(ang / 2)), np.cos(ang / 2)], dtype=np.float32)
        x += half * rng.uniform(0.3, 1.2)
        pose.append((q, tc.placed(com, q, [x, rng.uniform(-2.0, 2.0), rng.uniform(-2.0, 2.0)]), com))
        x += half * rng.uniform(0.3, 1.2)
    resp = (0.25, 0.6, 0.4)
    # ---- a collidable per body
    qs = many.collidable_queries(n)
    want, o_want = [], []  # the single-object calls' lists and the ORACLE's (the batched forms share device code with the former)
    for k, g in enumerate(objs):
        q, t, com = pose[k]
        mode = int(rng.integers(0, 3))
        qs[k]["mode"], qs[k]["rotation_xyzw"], qs[k]["translation"], qs[k]["response"] = mode, q, t, resp
        qs[k]["collidable_id_a"], qs[k]["collidable_id_b"], qs[k]["body_a"], qs[k]["body_b"] = 40 + k, 7, k, 0x80000000
        centre = np.array([float(rng.uniform(-5, x + 5)), float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8))], dtype=np.float32)
        if mode == 0:
            r = float(rng.uniform(2.0, 14.0))
            qs[k]["shape3"], qs[k]["shape1"] = centre, r
            want.append(g.sphere_contacts(q, t, centre, r, 40 + k, 7, k, 0x80000000, resp, capacity=65536))
            o_want.append(tcon.oracle_contact_list(orcs[k], q, t, centre, r, 40 + k, 7, k, 0x80000000, resp))
        elif mode == 1:
            nrm = rng.normal(size=3)
            nrm = (nrm / np.linalg.norm(nrm)).astype(np.float32)
            d = float(rng.uniform(-6.0, 6.0))
            qs[k]["shape3"], qs[k]["shape1"] = nrm, d
            want.append(g.plane_contacts(q, t, nrm, d, 40 + k, 7, k, 0x80000000, resp, capacity=65536))
            o_want.append(tcon.oracle_plane_contact_list(orcs[k], q, t, nrm, d, 40 + k, 7, k, 0x80000000, resp))
        else:
            v = (rng.normal(size=3) * rng.uniform(0.0, 12.0)).astype(np.float32)
            r = float(rng.uniform(1.5, 8.0))
            qs[k]["shape3"], qs[k]["shape3b"], qs[k]["shape1"] = centre, v, r
            want.append(g.capsule_contacts(q, t, centre, v, r, 40 + k, 7, k, 0x80000000, resp, capacity=65536))
            o_want.append(tcon.oracle_capsule_contact_list(orcs[k], q, t, centre, v, r, 40 + k, 7, k, 0x80000000, resp))
    got, off = many.voxel_object_contacts_many(objs, qs)
    assert off[-1] == len(got) == sum(len(w) for w in want)
    for k, w in enumerate(want):
        assert got[off[k]:off[k + 1]].tobytes() == w.tobytes(), f"collidable of body {k}"
        tcon.assert_contacts_equal(got[off[k]:off[k + 1]], o_want[k])
    exercised = [len(got)]
    # ---- a random list of pairs
    pair_ids = [(int(a), int(b)) for a, b in rng.integers(0, n, (int(rng.integers(1, 10)), 2)) if a != b]
    pairs, want = [], []
    for i, j in pair_ids:
        (qa, ta, ca), (qb, tb, cb) = pose[i], pose[j]
        pairs.append(dict(a=objs[i], b=objs[j], rotation_a=qa, translation_a=ta, center_of_mass_a=ca, rotation_b=qb, translation_b=tb, center_of_mass_b=cb,
                          collidable_id_a=100 + i, collidable_id_b=100 + j, body_a=i, body_b=j, response=resp))
        want.append(objs[i].mutual_contacts(qa, ta, ca, objs[j], qb, tb, cb, 100 + i, 100 + j, i, j, resp))
    got, off = many.mutual_voxel_object_contacts_many(many.mutual_queries(pairs))
    assert off[-1] == len(got) == sum(len(w) for w in want)
    for k, w in enumerate(want):
        assert got[off[k]:off[k + 1]].tobytes() == w.tobytes(), f"pair {pair_ids[k]}"
        i, j = pair_ids[k]
        (qa, ta, ca), (qb, tb, cb) = pose[i], pose[j]
        ow, _ = tc.oracle_contact_list(orcs[i], o_probes[i], ca, qa, ta, orcs[j], o_probes[j], cb, qb, tb, 100 + i, 100 + j, i, j, resp)
        tc.assert_contacts_equal(got[off[k]:off[k + 1]], ow)
    exercised.append(len(got))
    # ---- a bite per body, mesh and probes of all synced in one call each; twins through the single-object calls
    twins = [tc.both(ctx, g_, e_)[1] for g_, e_ in shapes]
    for t in twins:
        t.collision_probes_recompute()
    inv = []
    for g, t in zip(objs, twins):
        occ = np.asarray(g.update_occupied_voxel_ranges(), dtype=np.float64)
        c = (occ[:, 0] + rng.uniform(0.0, 1.0, 3) * (occ[:, 1] - occ[:, 0])).astype(np.float32)
        r = float(rng.uniform(1.5, 6.0))
        rg, rt = g.absorb_sphere(c, r + 2.0, r), t.absorb_sphere(c, r + 2.0, r)
        np.testing.assert_array_equal(rg["invalidated"], rt["invalidated"])
        t.mesh.sync_with_voxel_object(rt["invalidated"])
        t.collision_probes_sync(rt["invalidated"])
        inv.append(rg["invalidated"])
    many.mesh_sync_many([g.mesh for g in objs], inv)
    ns = many.collision_probes_sync_many(objs, inv)
    for k, (g, t) in enumerate(zip(objs, twins)):
        (gp, ge), (tp, te) = g.collision_probes(), t.collision_probes()
        assert int(ns[k]) == len(tp) == len(gp), k
        np.testing.assert_array_equal(ge, te)
        for e in te:
            np.testing.assert_array_equal(gp[e[3]:e[4]].view(np.uint32), tp[e[3]:e[4]].view(np.uint32))
    exercised.append(int(sum(int(np.asarray(i_).sum()) for i_ in inv)))
    assert pu.fuzzing() or (exercised[0] > 100 and exercised[1] > 0 and exercised[2] > 5), exercised  # (the committed seeds are ones that meet something)
    for g in objs + twins:
        g.close()
