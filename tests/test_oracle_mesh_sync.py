"""The oracle's incremental remesh (VoxelObjectMesh::sync_with_voxel_object, mesh.rs:355-456 over the ChunkSubmeshManager, mesh.rs:699-849,
and RangeAllocator, impact_containers/src/range_allocator.rs): the reference's own RangeAllocator tests re-typed as known answers, and the
synced mesh against a full rebuild of the edited object, chunk by chunk."""
import numpy as np
import pytest

import oracle_lib as ol
from impact_amd import scenes


def test_range_allocator_known_answers():
    """impact_containers/src/range_allocator.rs:140-262, every test of the reference"""
    R = ol.range_allocator_script
    assert R([("alloc", 1)]) == [None]  # allocates_nothing_before_freed
    assert R([("free", 2, 6), ("alloc", 4), ("alloc", 1)]) == [None, (2, 6), None]  # frees_and_allocates_single_range
    assert R([("free", 2, 6), ("free", 10, 12), ("alloc", 2), ("alloc", 4)])[2:] == [(10, 12), (2, 6)]  # allocates_range_in_smallest_slot
    assert R([("free", 2, 12), ("alloc", 4), ("alloc", 4), ("alloc", 4), ("alloc", 2), ("alloc", 1)])[1:] == [(2, 6), (6, 10), None, (10, 12), None]
    assert R([("free", 2, 5), ("free", 6, 9), ("merge",), ("alloc", 6)])[3] is None  # does_not_merge_two_disconnected_free_ranges
    assert R([("free", 2, 6), ("free", 6, 8), ("merge",), ("alloc", 6), ("alloc", 1)])[3:] == [(2, 8), None]
    assert R([("free", 2, 6), ("free", 6, 8), ("free", 8, 42), ("merge",), ("alloc", 40), ("alloc", 1)])[4:] == [(2, 42), None]
    assert R([("free", 2, 6), ("free", 6, 8), ("free", 8, 42), ("free", 42, 50), ("merge",), ("alloc", 48), ("alloc", 1)])[5:] == [(2, 50), None]
    # unmerged neighbours do not serve a request larger than either; two separate runs merge separately
    assert R([("free", 2, 6), ("free", 6, 8), ("alloc", 6)])[2] is None
    assert R([("free", 0, 2), ("free", 2, 4), ("free", 10, 11), ("free", 11, 14), ("merge",), ("alloc", 4), ("alloc", 4)])[5:] == [(0, 4), (10, 14)]


def per_chunk(mesh):
    """chunk -> (positions, normals, chunk-local indices, index materials, obscured table) of every submesh"""
    out = {}
    for sm in mesh.submeshes:
        ioff, icnt, voff, vcnt = int(sm[3]), int(sm[4]), int(sm[13]), int(sm[14])
        idx = mesh.indices[ioff:ioff + icnt].astype(np.int64) - voff
        assert idx.min() >= 0 and idx.max() < vcnt
        out[tuple(int(x) for x in sm[:3])] = (mesh.positions[voff:voff + vcnt].tobytes(), mesh.normals[voff:voff + vcnt].tobytes(), idx.tobytes(),
                                              mesh.index_materials[ioff:ioff + icnt].tobytes(), sm[5:13].tobytes())
    assert len(out) == len(mesh.submeshes)
    return out


def assert_ranges_disjoint(mesh):
    for col_off, col_cnt, total in ((13, 14, len(mesh.positions)), (3, 4, len(mesh.indices))):
        r = sorted((int(sm[col_off]), int(sm[col_off]) + int(sm[col_cnt])) for sm in mesh.submeshes)
        assert all(a[1] <= b[0] for a, b in zip(r, r[1:])) and (not r or r[-1][1] <= total)


@pytest.mark.parametrize("case", ["bite", "cavity_then_bite", "cut_through"])
def test_synced_mesh_equals_full_rebuild_chunk_by_chunk(case):
    o = ol.OracleObject.from_sdf(scenes.sphere_scene(30.0), 1.0, 0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    m = ol.OracleMeshHandle(o)
    before = m.get()
    ctr = np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)
    edits = {"bite": [(ctr + np.float32(30.0) * np.array([0.6, 0.0, 0.8], np.float32), 11.0)],
             "cavity_then_bite": [(ctr + np.array([1.0, 2.0, -3.0], np.float32), 9.0), (ctr + np.array([0.0, 30.0, 0.0], np.float32), 8.0)],
             "cut_through": [(ctr + np.array([0.0, float(y), 0.0], np.float32), 12.0) for y in (-24, -8, 8, 24)]}[case]
    for c, r in edits:
        res = o.absorb_sphere(c, r + 2.0, r)
        m.sync(res["invalidated"])
        got = m.get()
        want = o.mesh()  # full rebuild
        assert per_chunk(got) == per_chunk(want)
        assert_ranges_disjoint(got)
        assert len(got.positions) >= len(want.positions)  # buffers only grow; freed ranges may stay unused
    after = m.get()
    assert len(after.positions) >= len(before.positions)
    untouched = set(per_chunk(before).items()) & set(per_chunk(after).items())
    assert len(untouched) > 10  # chunks away from the edits keep their data


def test_reuse_of_freed_ranges_and_appending():
    """write_chunk frees a chunk's old ranges before allocating (mesh.rs:751-809): re-meshed chunks that fit a freed range land inside
    the old buffers, those that outgrew every free range are appended; chunks that were not invalidated keep their ranges"""
    o = ol.OracleObject.from_sdf(scenes.sphere_scene(30.0), 1.0, 0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    m = ol.OracleMeshHandle(o)
    before = m.get()
    ctr = np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)
    res = o.absorb_sphere(ctr + np.float32(30.0) * np.array([0.0, 0.0, 1.0], np.float32), 9.0, 7.0)
    m.sync(res["invalidated"])
    after = m.get()
    inval = res["invalidated"].reshape(o.chunk_counts)
    b = {tuple(int(x) for x in sm[:3]): sm for sm in before.submeshes}
    a = {tuple(int(x) for x in sm[:3]): sm for sm in after.submeshes}
    remeshed = [k for k in a if inval[k]]
    assert all(int(a[k][13]) >= len(before.positions) for k in remeshed)  # every touched chunk grew: all appended, their old ranges are free now
    for k in a:
        if not inval[k]:
            np.testing.assert_array_equal(a[k], b[k])
    assert len(after.positions) > len(before.positions)
    # a second, deeper bite at the same place: the chunks lose surface, and what they need now fits the ranges freed before
    res2 = o.absorb_sphere(ctr + np.float32(30.0) * np.array([0.0, 0.0, 1.0], np.float32), 22.0, 20.0)
    m.sync(res2["invalidated"])
    third = m.get()
    inval2 = res2["invalidated"].reshape(o.chunk_counts)
    c = {tuple(int(x) for x in sm[:3]): sm for sm in third.submeshes}
    assert any(inval2[k] and int(c[k][13]) < len(before.positions) for k in c)  # landed in a range freed by the first sync
    assert per_chunk(third) == per_chunk(o.mesh())
    assert_ranges_disjoint(third)


def test_synced_probes_equal_recomputed_probes_chunk_by_chunk():
    """sync_with_voxel_object_and_mesh (collidable.rs:394-433, 524-612) after each mesh sync: every chunk's probe points equal those of a
    recompute over the synced mesh; ranges are disjoint; chunks that were not invalidated keep their ranges"""
    o = ol.OracleObject.from_sdf(scenes.sphere_scene(30.0), 1.0, 0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    m = ol.OracleMeshHandle(o)
    probes = ol.OracleProbes(m)
    p0, e0 = probes.get()
    assert len(e0) > 20 and int(e0[-1][4]) == len(p0)
    ctr = np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)
    top = ctr + np.float32(30.0) * np.array([0.0, 0.0, 1.0], np.float32)
    prev = {tuple(int(x) for x in e[:3]): (int(e[3]), int(e[4])) for e in e0}
    for r in (7.0, 20.0, 31.0):
        res = o.absorb_sphere(top, r + 2.0, r)
        m.sync(res["invalidated"])
        probes.sync(res["invalidated"])
        pts, ent = probes.get()
        fresh_pts, fresh_ent = ol.OracleProbes(m).get()  # recompute over the same (synced) mesh: per-chunk ground truth
        got = {tuple(int(x) for x in e[:3]): pts[int(e[3]):int(e[4])].tobytes() for e in ent}
        want = {tuple(int(x) for x in e[:3]): fresh_pts[int(e[3]):int(e[4])].tobytes() for e in fresh_ent}
        assert got == want
        spans = sorted((int(e[3]), int(e[4])) for e in ent)
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= len(pts)
        inval = res["invalidated"].reshape(o.chunk_counts)
        now = {tuple(int(x) for x in e[:3]): (int(e[3]), int(e[4])) for e in ent}
        for k, span in now.items():
            if not inval[k] and k in prev:
                assert span == prev[k]
        prev = now
    assert len(pts) >= len(p0)
