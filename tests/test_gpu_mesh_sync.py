"""GPU parity of the incremental remesh (row a7: VoxelObjectMesh::sync_with_voxel_object, mesh.rs:355-456): after every edit the synced
mesh — submesh table in slot order, vertex / index ranges chosen by the best-fit range allocators, the data inside every live range —
equals the oracle's (tests/test_oracle_mesh_sync.py pins the oracle), and equals a full rebuild chunk by chunk."""
import numpy as np
import pytest

import oracle_lib as ol
import parity_util as pu
from impact_amd import capi, scenes
from impact_amd.voxel import VoxelObjectMesh

pytestmark = pytest.mark.gpu


def both(ctx, graph, extent=1.0):
    o = pu.oracle_from_graph(graph, extent)
    g = pu.gpu_from_graph(ctx, graph, extent)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    g.compute_all_derived_state()
    g.update_occupied_voxel_ranges()
    g.label_regions()
    return o, g


def assert_synced_meshes_equal(om: ol.OracleMesh, gm):
    pos, nrm, idx, im, sub = gm
    assert len(sub) == len(om.submeshes) and len(pos) == len(om.positions) and len(idx) == len(om.indices)
    want = om.submeshes
    for f, col in (("index_offset", 3), ("index_count", 4), ("vertex_offset", 13), ("vertex_count", 14)):
        np.testing.assert_array_equal(sub[f], want[:, col], err_msg=f)
    np.testing.assert_array_equal(sub["chunk_indices"], want[:, :3])
    np.testing.assert_array_equal(sub["is_obscured_from_direction"].reshape(len(sub), 8), want[:, 5:13])
    for sm in want:  # the data of every live range, bit for bit
        ioff, icnt, voff, vcnt = int(sm[3]), int(sm[4]), int(sm[13]), int(sm[14])
        np.testing.assert_array_equal(pos[voff:voff + vcnt].view(np.uint32), om.positions[voff:voff + vcnt].view(np.uint32))
        np.testing.assert_array_equal(nrm[voff:voff + vcnt].view(np.uint32), om.normals[voff:voff + vcnt].view(np.uint32))
        np.testing.assert_array_equal(idx[ioff:ioff + icnt], om.indices[ioff:ioff + icnt])
        np.testing.assert_array_equal(im[ioff:ioff + icnt], om.index_materials[ioff:ioff + icnt])


@pytest.mark.parametrize("case", ["bites", "cut_through", "capsule_then_sphere", "eaten_whole"])
def test_sync_after_edits(ctx, case):
    o, g = both(ctx, scenes.sphere_scene(30.0))
    om = ol.OracleMeshHandle(o)
    gm = VoxelObjectMesh.create(g)
    ctr = np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)
    top = ctr + np.float32(30.0) * np.array([0.0, 0.0, 1.0], np.float32)
    edits = {
        "bites": [("s", top, 7.0), ("s", top, 20.0), ("s", ctr + np.float32(30.0) * np.array([0.6, 0.0, 0.8], np.float32), 11.0)],
        "cut_through": [("s", ctr + np.array([0.0, float(y), 0.0], np.float32), 12.0) for y in (-24, -8, 8, 24)],
        "capsule_then_sphere": [("c", ctr + np.array([-40.0, 3.0, 20.0], np.float32), np.array([80.0, -6.0, 4.0], np.float32), 6.0), ("s", top, 9.0)],
        "eaten_whole": [("s", ctr, 50.0)],
    }[case]
    for e in edits:
        if e[0] == "s":
            ro = o.absorb_sphere(e[1], e[2] + 2.0, e[2])
            rg = g.absorb_sphere(e[1], e[2] + 2.0, e[2])
        else:
            ro = o.absorb_capsule(e[1], e[2], e[3] + 2.0, e[3])
            rg = g.absorb_capsule(e[1], e[2], e[3] + 2.0, e[3])
        np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
        om.sync(ro["invalidated"])
        gm.sync_with_voxel_object(rg["invalidated"])
        assert_synced_meshes_equal(om.get(), gm.download())
        # what the renderer is told to re-upload (VoxelMeshModifications, mesh.rs:113-123)
        want_ranges, want_removed = om.modifications()
        got_ranges, got_removed = gm.mesh_modifications()
        np.testing.assert_array_equal(got_ranges, want_ranges)
        assert got_removed == want_removed
        if e is edits[0]:
            assert len(got_ranges) > 0 or case == "eaten_whole"
        if e is not edits[-1] and len(edits) > 2:  # sometimes report in between, sometimes let the ranges pile up
            om.report_synchronized()
            gm.report_gpu_resources_synchronized()
    if case == "eaten_whole":
        assert gm.mesh_modifications()[1]  # every chunk lost its submesh
        assert gm.n_chunks() == 0
    else:
        # and a full rebuild afterwards gives the same submeshes, chunk by chunk
        synced = {tuple(s["chunk_indices"]): s for s in gm.download()[4]}
        pos, nrm, idx, im, _ = gm.download()
        full = VoxelObjectMesh.create(g)
        fpos, fnrm, fidx, fim, fsub = full.download()
        assert len(fsub) == len(synced)
        for s in fsub:
            t = synced[tuple(s["chunk_indices"])]
            assert int(t["vertex_count"]) == int(s["vertex_count"]) and int(t["index_count"]) == int(s["index_count"])
            a0, a1, b0, b1 = int(t["vertex_offset"]), int(t["index_offset"]), int(s["vertex_offset"]), int(s["index_offset"])
            nv, ni = int(s["vertex_count"]), int(s["index_count"])
            np.testing.assert_array_equal(pos[a0:a0 + nv].view(np.uint32), fpos[b0:b0 + nv].view(np.uint32))
            np.testing.assert_array_equal(idx[a1:a1 + ni].astype(np.int64) - a0, fidx[b1:b1 + ni].astype(np.int64) - b0)
    g.close()


def test_sync_after_full_remesh_restarts_the_bookkeeping(ctx):
    """a full ivx_remesh between syncs drops the free ranges (recreate -> clear, mesh.rs:286-300)"""
    o, g = both(ctx, scenes.sphere_scene(24.0))
    gm = VoxelObjectMesh.create(g)
    ctr = np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)
    for r in (6.0, 12.0):
        c = ctr + np.float32(24.0) * np.array([1.0, 0.0, 0.0], np.float32)
        ro, rg = o.absorb_sphere(c, r + 2.0, r), g.absorb_sphere(c, r + 2.0, r)
        gm.sync_with_voxel_object(rg["invalidated"])
    gm.recreate()
    om = ol.OracleMeshHandle(o)  # the oracle mesh recreated at the same point
    c = ctr + np.float32(24.0) * np.array([0.0, 1.0, 0.0], np.float32)
    ro, rg = o.absorb_sphere(c, 9.0, 7.0), g.absorb_sphere(c, 9.0, 7.0)
    om.sync(ro["invalidated"])
    gm.sync_with_voxel_object(rg["invalidated"])
    assert_synced_meshes_equal(om.get(), gm.download())
    g.close()


def test_probes_follow_the_incremental_remesh_and_feed_the_mutual_contacts(ctx):
    """MeshedVoxelObject::sync_mesh_with_object in full: absorb -> mesh sync -> probe sync (collidable.rs:394-433), three times over; the
    entries (chunk, range) and every live point equal the oracle's, and the contacts against a second body computed from the synced
    probes equal the oracle's contacts from its synced probes"""
    import test_gpu_collide as tc

    o, g = both(ctx, scenes.sphere_scene(30.0))
    b_o, b_g = both(ctx, scenes.sphere_scene(18.0))
    om, gm = ol.OracleMeshHandle(o), VoxelObjectMesh.create(g)
    VoxelObjectMesh.create(b_g)
    op = ol.OracleProbes(om)
    g.collision_probes_recompute()
    b_g.collision_probes_recompute()
    pb = b_o.collision_probes(b_o.mesh())
    ctr = np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)
    cb = np.array([0.5 * (a + b) for a, b in b_o.info()["occupied_voxel_ranges"]], dtype=np.float32)
    top = ctr + np.float32(30.0) * np.array([0.0, 0.0, 1.0], np.float32)
    ident = np.array([0, 0, 0, 1], np.float32)
    tb = (cb.astype(np.float64) - np.array([0.0, 3.0, 41.0])).astype(np.float32)  # B sits on top of A, where the bites happen
    n_contacts = []
    for r in (7.0, 16.0, 24.0):
        ro, rg = o.absorb_sphere(top, r + 2.0, r), g.absorb_sphere(top, r + 2.0, r)
        om.sync(ro["invalidated"])
        op.sync(ro["invalidated"])
        gm.sync_with_voxel_object(rg["invalidated"])
        n = g.collision_probes_sync(rg["invalidated"])
        want_pts, want_ent = op.get()
        got_pts, got_ent = g.collision_probes()
        assert n == len(want_pts) == len(got_pts)
        np.testing.assert_array_equal(got_ent, want_ent)
        for e in want_ent:
            np.testing.assert_array_equal(got_pts[e[3]:e[4]].view(np.uint32), want_pts[e[3]:e[4]].view(np.uint32))
        want, wi = tc.oracle_contact_list(o, (want_pts, want_ent), ctr, ident, ctr, b_o, pb, cb, ident, tb, 3, 4, 0, 1, (0.2, 0.5, 0.4))
        got = g.mutual_contacts(ident, ctr, ctr, b_g, ident, tb, cb, 3, 4, 0, 1, (0.2, 0.5, 0.4))
        tc.assert_contacts_equal(got, want)
        n_contacts.append(len(want))
    assert len(want_ent) > 10 and max(n_contacts) > 5, n_contacts
    g.close()
    b_g.close()


def test_probes_of_many_objects_follow_their_meshes(ctx):
    """ivx_collision_probes_sync_many: four objects of different sizes (block sizes 8 and 4, one with buffers that must grow, one whose round
    invalidates nothing), three rounds of absorb -> mesh sync -> probe sync for all in one call; entries and live points of every object equal
    the oracle's after every round, and equal what the single-object call leaves on a twin of the object"""
    from impact_amd import many

    specs = [(scenes.sphere_scene(30.0), 1.0), (scenes.box_scene((9.0, 12.0, 20.0)), 1.0), (scenes.asteroid_scene(0.4), 1.0), (scenes.sphere_scene(20.0), 1.0)]
    objs = [both(ctx, gr, ex) for gr, ex in specs]
    twins = [both(ctx, gr, ex)[1] for gr, ex in specs]
    oms = [ol.OracleMeshHandle(o) for o, _ in objs]
    ops = [ol.OracleProbes(om) for om in oms]
    gms = [VoxelObjectMesh.create(g) for _, g in objs]
    tms = [VoxelObjectMesh.create(t) for t in twins]
    for (_, g), t in zip(objs, twins):
        g.collision_probes_recompute()
        t.collision_probes_recompute()
    ctrs = [np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32) for o, _ in objs]
    half = [np.array([0.5 * (b - a) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32) for o, _ in objs]
    for rnd, (frac, r) in enumerate([(1.0, 5.0), (0.8, 9.0), (0.6, 4.0)]):
        inv = []
        for i, ((o, g), t) in enumerate(zip(objs, twins)):
            c = ctrs[i] + np.float32(frac) * half[i] * np.array([0.0, 0.0, 1.0], np.float32)
            rr = np.float32(r if i != 1 else 2.0)
            if i == 3 and rnd == 1:
                c = ctrs[i] + np.float32(100.0)  # far away: nothing invalidated this round
            ro, rg, rt = o.absorb_sphere(c, rr + 2.0, rr), g.absorb_sphere(c, rr + 2.0, rr), t.absorb_sphere(c, rr + 2.0, rr)
            np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
            oms[i].sync(ro["invalidated"])
            ops[i].sync(ro["invalidated"])
            tms[i].sync_with_voxel_object(rt["invalidated"])
            t.collision_probes_sync(rt["invalidated"])
            inv.append(rg["invalidated"])
        many.mesh_sync_many(gms, inv)
        st0 = np.zeros(3, dtype=np.uint64)
        st1 = np.zeros(3, dtype=np.uint64)
        capi.check(capi.lib().ivx_many_stats(ctx.h, capi.ptr(st0)))
        ns = many.collision_probes_sync_many([g for _, g in objs], inv)
        capi.check(capi.lib().ivx_many_stats(ctx.h, capi.ptr(st1)))
        rec, iss = int(st1[0] - st0[0]), int(st1[1] - st0[1])
        assert rec >= 3 * iss // 2 and iss <= 9, (rec, iss)  # (upload | select x2 merged over the objects, then fills | upload | gather)
        for i, ((o, g), t) in enumerate(zip(objs, twins)):
            want_pts, want_ent = ops[i].get()
            got_pts, got_ent = g.collision_probes()
            assert int(ns[i]) == len(want_pts) == len(got_pts), (rnd, i)
            np.testing.assert_array_equal(got_ent, want_ent)
            for e in want_ent:
                np.testing.assert_array_equal(got_pts[e[3]:e[4]].view(np.uint32), want_pts[e[3]:e[4]].view(np.uint32))
            tw_pts, tw_ent = t.collision_probes()
            np.testing.assert_array_equal(got_ent, tw_ent)
            assert len(tw_pts) == len(got_pts)
    assert sum(len(ops[i].get()[1]) for i in range(4)) > 30
    # the empty list, and an object listed twice
    from impact_amd.capi import IvxError

    assert len(many.collision_probes_sync_many([], [])) == 0
    with pytest.raises(IvxError):
        many.collision_probes_sync_many([objs[0][1], objs[0][1]], [inv[0], inv[0]])
    for (_, g), t in zip(objs, twins):
        g.close()
        t.close()


def test_edit_and_sync_in_two_halves(ctx):
    """ivx_absorb_sphere_enqueue / ivx_absorb_collect and ivx_mesh_sync_enqueue / _collect: the edit's kernels, the sweep over the touched
    chunks and their neighbours, the region resolve and the count of what the invalidated meshes need on the stream behind ONE wait; the sync
    placed from those sizes without a count pass or a read-back of its own. Same results as the oracle's, edit after edit."""
    o, g = both(ctx, scenes.asteroid_scene(0.5))
    om = ol.OracleMeshHandle(o)
    gm = VoxelObjectMesh.create(g)
    ctr = np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)
    for step, (d, r) in enumerate([((0.0, 0.0, 48.0), 14.0), ((30.0, 5.0, 30.0), 9.0), ((0.0, 0.0, 40.0), 22.0), ((-50.0, 0.0, 0.0), 30.0)]):
        c = ctr + np.asarray(d, dtype=np.float32)
        ro = o.absorb_sphere(c, r + 2.0, r)
        g.absorb_sphere_enqueue(c, r + 2.0, r)
        rg = g.absorb_collect()
        np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
        np.testing.assert_array_equal(rg["emptied_by_type"], ro["emptied_by_type"])
        assert (rg["touched_chunks"], rg["removed_chunks"]) == (ro["touched_chunks"], ro["removed_chunks"])
        scale = np.maximum(np.abs(ro["removed64"]), 1e-300)
        assert np.all(np.abs(rg["removed_moments"] - ro["removed64"]) <= 1e-5 * scale + 1e-9)
        pu.assert_edited_objects_equal(o, g, f"edit {step}: ", with_mesh=False)
        om.sync(ro["invalidated"])
        gm.sync_enqueue(rg["invalidated"])
        gm.sync_collect()
        assert_synced_meshes_equal(om.get(), gm.download())
    with pytest.raises(capi.IvxError):
        g.absorb_collect()  # nothing in flight
    g.close()


def test_edit_and_sync_overlapped(ctx):
    """ivx_absorb_sphere_enqueue -> ivx_mesh_sync_enqueue(NULL) -> ivx_absorb_collect -> ivx_mesh_sync_collect: the sync placed from the mesh
    needs the edit's count role delivers early, its launches behind the edit's, while the edit is still in flight. Voxels, labels, edit results
    and the synced mesh equal the oracle's, edit after edit (bites of different sizes — the early records' place in the host-mapped block moves
    with the box —, one that touches nothing, one that removes chunks); a null set without an edit in flight is an error."""
    o, g = both(ctx, scenes.asteroid_scene(0.5))
    om = ol.OracleMeshHandle(o)
    gm = VoxelObjectMesh.create(g)
    ctr = np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)
    with pytest.raises(capi.IvxError):
        gm.sync_enqueue(None)  # no edit in flight
    c0 = ctr + np.asarray((0.0, 0.0, 52.0), dtype=np.float32)
    o.absorb_sphere(c0, 5.0, 3.0)
    g.absorb_sphere_enqueue(c0, 5.0, 3.0)
    with pytest.raises(capi.IvxError):
        gm.sync_enqueue(None)  # an edit in flight, but not one that delivers early
    r0 = g.absorb_collect()
    om.sync(r0["invalidated"])
    gm.sync_with_voxel_object(r0["invalidated"])
    g.set_early_mesh_needs(True)
    edits = [((0.0, 0.0, 48.0), 14.0), ((30.0, 5.0, 30.0), 9.0), ((0.0, 0.0, 40.0), 22.0), ((400.0, 0.0, 0.0), 5.0), ((-50.0, 0.0, 0.0), 30.0), ((0.0, 0.0, 44.0), 3.0),
             ((10.0, -20.0, 30.0), 26.0)]
    for step, (d, r) in enumerate(edits):
        c = ctr + np.asarray(d, dtype=np.float32)
        ro = o.absorb_sphere(c, r + 2.0, r)
        g.absorb_sphere_enqueue(c, r + 2.0, r)
        gm.sync_enqueue(None)
        rg = g.absorb_collect()
        gm.sync_collect()
        np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
        np.testing.assert_array_equal(rg["emptied_by_type"], ro["emptied_by_type"])
        assert (rg["touched_chunks"], rg["removed_chunks"]) == (ro["touched_chunks"], ro["removed_chunks"])
        pu.assert_edited_objects_equal(o, g, f"edit {step}: ", with_mesh=False)
        om.sync(ro["invalidated"])
        assert_synced_meshes_equal(om.get(), gm.download())
    assert sum(1 for d, r in edits if r > 20.0) >= 2
    # the usual order still works behind it (the early records are armed on every edit, taken or not)
    c = ctr + np.asarray((0.0, 25.0, 35.0), dtype=np.float32)
    ro, rg = o.absorb_sphere(c, 12.0, 10.0), g.absorb_sphere(c, 12.0, 10.0)
    om.sync(ro["invalidated"])
    gm.sync_with_voxel_object(rg["invalidated"])
    assert_synced_meshes_equal(om.get(), gm.download())
    pu.assert_edited_objects_equal(o, g, "after the overlapped edits: ")
    g.close()


def test_sync_of_several_edits_at_once(ctx):
    """two edits, then ONE sync over the union of what they invalidated (the reference syncs once per frame, lib.rs:729-733): the sizes of the
    first edit's chunks are not the last edit's — the sync counts the listed chunks itself"""
    o, g = both(ctx, scenes.sphere_scene(30.0))
    om = ol.OracleMeshHandle(o)
    gm = VoxelObjectMesh.create(g)
    ctr = np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)
    inv_o = np.zeros(g.n_chunks, dtype=bool)
    inv_g = np.zeros(g.n_chunks, dtype=bool)
    for d, r in (((0.0, 0.0, 30.0), 8.0), ((0.0, 30.0, 0.0), 10.0), ((4.0, 0.0, 26.0), 6.0)):
        c = ctr + np.asarray(d, dtype=np.float32)
        inv_o |= o.absorb_sphere(c, r + 2.0, r)["invalidated"]
        inv_g |= g.absorb_sphere(c, r + 2.0, r)["invalidated"]
    np.testing.assert_array_equal(inv_g, inv_o)
    om.sync(inv_o)
    gm.sync_with_voxel_object(inv_g)
    assert_synced_meshes_equal(om.get(), gm.download())
    pu.assert_edited_objects_equal(o, g, "after three edits: ")  # (a full remesh behind box sweeps: the active list is rebuilt first)
    g.close()
