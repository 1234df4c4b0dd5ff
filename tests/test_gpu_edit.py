"""GPU parity of the voxel edit op: an absorbing sphere eating into a voxel object — HIP path through the C ABI
against the oracle (tests/test_oracle_voxel.py pins the oracle's edit against an independent numpy restatement). After every
edit: voxel bytes, flags, chunk records, chunk-local labels bit-exact; regions equal after canonical relabelling; the mesh of
the edited object bit-exact; removed moments within 1e-5; emptied counts, touched / removed chunks and the set of invalidated
mesh chunks equal."""
import numpy as np
import pytest

import oracle_lib as ol
import parity_util as pu
from impact_amd import scenes
from impact_amd.voxel import VoxelObject

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


def absorb_both(o, g, center, radius, dens=None):
    ro = o.absorb_sphere(center, radius + 2.0, radius, dens)
    rg = g.absorb_sphere(center, radius + 2.0, radius, dens)
    assert rg["touched_chunks"] == ro["touched_chunks"] and rg["removed_chunks"] == ro["removed_chunks"]
    np.testing.assert_array_equal(rg["emptied_by_type"], ro["emptied_by_type"])
    np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
    scale = np.maximum(np.abs(ro["removed64"]), 1e-300)
    assert np.all(np.abs(rg["removed_moments"] - ro["removed64"]) <= 1e-5 * scale + 1e-9), (rg["removed_moments"], ro["removed64"])
    ro["regions"] = pu.assert_edited_objects_equal(o, g, densities=dens)
    return ro


def centre_of(o):
    return np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)


def test_bite_out_of_a_sphere_surface(ctx):
    o, g = both(ctx, scenes.sphere_scene(40.0))
    c = centre_of(o) + np.float32(40.0) * np.array([0.6, 0.0, 0.8], np.float32)
    r = absorb_both(o, g, c, 13.0)
    assert r["emptied_by_type"].sum() > 1000 and r["regions"] == 1
    g.close()


def test_cavity_inside_converts_uniform_chunks(ctx):
    """the sphere lies wholly inside the body: Uniform chunks of the touched box become NonUniform whether or not a voxel of
    them changes, a closed cavity appears"""
    o, g = both(ctx, scenes.sphere_scene(60.0))
    r = absorb_both(o, g, centre_of(o) + np.array([0.25, -0.5, 0.75], np.float32), 17.0)
    assert r["touched_chunks"] > 8 and r["regions"] == 1
    g.close()


def test_drilling_through_a_rod_splits_it(ctx):
    """successive absorbing spheres (the reference moves the absorber every frame) cut a rod in two: after the last bite the
    object has two regions and the smaller one splits off exactly as in the reference"""
    from impact_amd.sdf_graph import SDFGraph, SDFNode

    gr = SDFGraph()
    gr.add_node(SDFNode.new_box([120.0, 14.0, 14.0]))
    o, g = both(ctx, gr)
    c0 = centre_of(o)
    for t, off in enumerate((-6.0, 0.0, 6.0)):
        r = absorb_both(o, g, c0 + np.array([5.0, off, 0.5 * t], np.float32), 9.5)
    assert r["regions"] == 2
    rc_o, child_o, origin_o = o.split_off_smallest_region()
    rc_g, child_g, origin_g, _ = g.extract_any_disconnected_region()
    assert rc_o == 1 and rc_g == 1 and tuple(int(x) for x in origin_g) == tuple(origin_o)
    pu.assert_edited_objects_equal(o, g)
    pu.assert_edited_objects_equal(child_o, child_g)
    child_g.close()
    g.close()


def test_absorbing_everything_leaves_an_empty_object(ctx):
    o, g = both(ctx, scenes.sphere_scene(20.0))
    r = absorb_both(o, g, centre_of(o), 40.0)
    assert r["removed_chunks"] > 0
    assert g.count_regions() == 0
    g.close()


def test_sphere_that_misses_changes_nothing(ctx):
    o, g = both(ctx, scenes.sphere_scene(20.0))
    before = g.download()
    r = absorb_both(o, g, centre_of(o) + np.array([200.0, 0.0, 0.0], np.float32), 10.0)
    assert r["touched_chunks"] == 0
    after = g.download()
    for a, b in zip(before[:4], after[:4]):
        np.testing.assert_array_equal(a, b)
    g.close()


def test_multi_material_object_reports_absorbed_types(ctx):
    """dense upload with three voxel types and non-uniform densities: the tracker's per-type counts and the removed moments"""
    rng = np.random.default_rng(5)
    cc = (3, 3, 3)
    n = 48
    x, y, z = np.meshgrid(*[np.arange(n) + 0.5] * 3, indexing="ij")
    d = np.sqrt((x - 24) ** 2 + (y - 24) ** 2 + (z - 24) ** 2) - 19.0
    sd = np.clip(np.trunc(d.astype(np.float32) * np.float32(50.0)), -128, 127).astype(np.int8)
    ty = (np.floor(x / 16).astype(np.uint8) + np.floor(z / 24).astype(np.uint8)) % 3
    ty[sd >= 0] = 255
    sdt, tyt = ol.dense_to_tiled(sd), ol.dense_to_tiled(ty)
    o = ol.OracleObject.from_dense(cc, sdt, tyt)
    g = VoxelObject.from_dense(ctx, cc, sdt, tyt)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    g.compute_all_derived_state()
    g.update_occupied_voxel_ranges()
    g.label_regions()
    dens = rng.uniform(0.5, 3.0, 256).astype(np.float32)
    r = absorb_both(o, g, np.array([30.0, 22.5, 26.0], np.float32), 9.0, dens)
    assert np.count_nonzero(r["emptied_by_type"][:3]) >= 2
    g.close()


# ---- absorbing capsule ------------------------------------------------------------------------------------------------------------
def absorb_capsule_both(o, g, start, vec, radius, dens=None):
    ro = o.absorb_capsule(start, vec, radius + 2.0, radius, dens)
    rg = g.absorb_capsule(start, vec, radius + 2.0, radius, dens)
    assert rg["touched_chunks"] == ro["touched_chunks"] and rg["removed_chunks"] == ro["removed_chunks"]
    np.testing.assert_array_equal(rg["emptied_by_type"], ro["emptied_by_type"])
    np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
    scale = np.maximum(np.abs(ro["removed64"]), 1e-300)
    assert np.all(np.abs(rg["removed_moments"] - ro["removed64"]) <= 1e-5 * scale + 1e-9), (rg["removed_moments"], ro["removed64"])
    ro["regions"] = pu.assert_edited_objects_equal(o, g, densities=dens)
    return ro


@pytest.mark.parametrize("case", ["diagonal", "axis_aligned", "point", "tiny_component", "miss"])
def test_capsule_through_a_sphere(ctx, case):
    """a skew capsule through the body (every chunk clips the segment differently), an axis-aligned one (zero offset components
    take the slab test's other branch), a zero-length one (a sphere with <=), one with a 1e-9 component, and a miss"""
    o, g = both(ctx, scenes.sphere_scene(40.0))
    c = centre_of(o)
    if case == "diagonal":
        start, vec, r = c + np.array([-50.0, -37.5, -20.25], np.float32), np.array([101.0, 70.5, 44.0], np.float32), 6.0
    elif case == "axis_aligned":
        start, vec, r = c + np.array([0.5, -60.0, 35.0], np.float32), np.array([0.0, 120.0, 0.0], np.float32), 5.0
    elif case == "point":
        start, vec, r = c + np.array([30.0, 10.0, -12.0], np.float32), np.zeros(3, np.float32), 11.0
    elif case == "tiny_component":
        start, vec, r = c + np.array([-3.0, 2.0, -70.0], np.float32), np.array([1e-9, -1e-9, 140.0], np.float32), 9.0
    else:
        start, vec, r = c + np.array([90.0, 90.0, 0.0], np.float32), np.array([0.0, 0.0, 40.0], np.float32), 6.0
    res = absorb_capsule_both(o, g, start, vec, r)
    if case == "miss":
        assert res["touched_chunks"] == 0
    else:
        assert res["emptied_by_type"].sum() > 1000 and res["regions"] == 1
    g.close()


def test_capsule_across_the_reference_box_geometry(ctx):
    """modifying_voxels_within_capsule_finds_correct_voxels_across_chunks (object/intersection.rs:1301-1345): a box of 30x14x14
    voxels (extent 0.25, capsule scaled by 4 into voxel units) skewered along z by a capsule far longer than the object"""
    o, g = both(ctx, scenes.box_scene((30.0, 14.0, 14.0)), extent=0.25)
    res = absorb_capsule_both(o, g, np.array([15.2, 12.0, -200.0], np.float32), np.array([0.0, 0.0, 2000.0], np.float32), 4.0)
    assert res["emptied_by_type"].sum() > 500
    g.close()


def test_capsule_sweep_cuts_a_rod_in_two(ctx):
    """the swept absorber of a moving tool: one capsule across a rod severs it; the smaller part splits off as in the reference"""
    from impact_amd.sdf_graph import SDFGraph, SDFNode

    gr = SDFGraph()
    gr.add_node(SDFNode.new_box([120.0, 14.0, 14.0]))
    o, g = both(ctx, gr)
    c0 = centre_of(o)
    r = absorb_capsule_both(o, g, c0 + np.array([7.0, -14.0, -3.0], np.float32), np.array([2.0, 28.0, 6.0], np.float32), 10.5)
    assert r["regions"] == 2
    rc_o, child_o, origin_o = o.split_off_smallest_region()
    rc_g, child_g, origin_g, _ = g.extract_any_disconnected_region()
    assert rc_o == 1 and rc_g == 1 and tuple(int(x) for x in origin_g) == tuple(origin_o)
    pu.assert_edited_objects_equal(o, g)
    pu.assert_edited_objects_equal(child_o, child_g)
    child_g.close()
    g.close()


def test_capsule_on_multi_material_object(ctx):
    rng = np.random.default_rng(9)
    cc = (3, 3, 3)
    n = 48
    x, y, z = np.meshgrid(*[np.arange(n) + 0.5] * 3, indexing="ij")
    d = np.sqrt((x - 24) ** 2 + (y - 24) ** 2 + (z - 24) ** 2) - 19.0
    sd = np.clip(np.trunc(d.astype(np.float32) * np.float32(50.0)), -128, 127).astype(np.int8)
    ty = (np.floor(y / 16).astype(np.uint8) + np.floor(x / 24).astype(np.uint8)) % 3
    ty[sd >= 0] = 255
    sdt, tyt = ol.dense_to_tiled(sd), ol.dense_to_tiled(ty)
    o = ol.OracleObject.from_dense(cc, sdt, tyt)
    g = VoxelObject.from_dense(ctx, cc, sdt, tyt)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    g.compute_all_derived_state()
    g.update_occupied_voxel_ranges()
    g.label_regions()
    dens = rng.uniform(0.5, 3.0, 256).astype(np.float32)
    r = absorb_capsule_both(o, g, np.array([6.0, 10.5, 20.0], np.float32), np.array([36.0, 27.0, 9.5], np.float32), 5.0, dens)
    assert np.count_nonzero(r["emptied_by_type"][:3]) >= 2
    g.close()


# ---- mutual absorption ----------------------------------------------------------------------------------------------------------------
def rot64(q, v):
    x, y, z, w = [float(a) for a in q]
    b = np.array([x, y, z])
    return v * (w * w - b @ b) + b * (2 * (v @ b)) + np.cross(b, v) * (2 * w)


def absorb_mutual_both(A, GA, qa, ta, B, GB, qb, tb, smooth, dens_a=None, dens_b=None):
    roa, rob = A.absorb_mutual(qa, ta, B, qb, tb, smooth, dens_a, dens_b)
    rga, rgb = GA.absorb_mutual(qa, ta, GB, qb, tb, smooth, dens_a, dens_b)
    for ro, rg, o, g, dens in ((roa, rga, A, GA, dens_a), (rob, rgb, B, GB, dens_b)):
        assert rg["emptied_voxels"] == ro["emptied_voxels"]
        assert rg["touched_chunks"] == ro["touched_chunks"] and rg["removed_chunks"] == ro["removed_chunks"]
        np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
        scale = np.maximum(np.abs(ro["removed64"]), 1e-300)
        assert np.all(np.abs(rg["removed_moments"] - ro["removed64"]) <= 1e-5 * scale + 1e-9), (rg["removed_moments"], ro["removed64"])
        ro["regions"] = pu.assert_edited_objects_equal(o, g, densities=dens)
    return roa, rob


@pytest.mark.parametrize("smooth", [0.0, 2.0])
@pytest.mark.parametrize("case", ["aligned", "rotated_mixed_extents", "small_inside_big", "apart"])
def test_mutual_absorption(ctx, case, smooth):
    """apply_mutual_absorption: two spheres side by side, a rotated box of another voxel extent pushed into a sphere (the snapshot
    padding is two A voxels), a small body wholly inside a big one (it disappears; the big one gets a cavity), and two bodies whose
    boxes do not meet"""
    if case == "aligned":
        (A, GA), (B, GB) = both(ctx, scenes.sphere_scene(30.0)), both(ctx, scenes.sphere_scene(22.0))
        qa, qb, centre_b = (0, 0, 0, 1), (0, 0, 0, 1), [0.25, 44.0, -0.5]
    elif case == "rotated_mixed_extents":
        (A, GA), (B, GB) = both(ctx, scenes.sphere_scene(40.0), 0.5), both(ctx, scenes.box_scene((22.0, 18.0, 30.0)), 1.0)
        ax = np.array([0.2, -0.4, 1.0]) / np.linalg.norm([0.2, -0.4, 1.0])
        qa = (np.sin(0.15), 0.0, 0.0, np.cos(0.15))
        qb, centre_b = (*(ax * np.sin(0.4)), np.cos(0.4)), [1.0, 25.0, 2.0]
    elif case == "small_inside_big":
        (A, GA), (B, GB) = both(ctx, scenes.sphere_scene(40.0)), both(ctx, scenes.sphere_scene(9.0))
        qa, qb, centre_b = (0, 0, 0, 1), (0.0, np.sin(0.3), 0.0, np.cos(0.3)), [3.0, -2.0, 5.0]
    else:
        (A, GA), (B, GB) = both(ctx, scenes.sphere_scene(20.0)), both(ctx, scenes.sphere_scene(20.0))
        qa, qb, centre_b = (0, 0, 0, 1), (0, 0, 0, 1), [0.0, 90.0, 0.0]
    qa, qb = np.asarray(qa, np.float32), np.asarray(qb, np.float32)
    ca = centre_of(A).astype(np.float64) * (0.5 if case == "rotated_mixed_extents" else 1.0)
    cb = centre_of(B).astype(np.float64)
    ta = (ca - rot64(qa, np.zeros(3))).astype(np.float32)  # A's centre at the world origin
    tb = (cb - rot64(qb, np.asarray(centre_b, dtype=np.float64))).astype(np.float32)
    ra, rb = absorb_mutual_both(A, GA, qa, ta, B, GB, qb, tb, smooth)
    if case == "apart":
        assert ra["touched_chunks"] == 0 and rb["touched_chunks"] == 0
    else:
        assert ra["emptied_voxels"] > 500 and rb["emptied_voxels"] > 500
    if case == "small_inside_big":
        assert GB.count_regions() == 0  # (emptied, not void: the distances just outside the new surface are small)
    GA.close()
    GB.close()


def test_mutual_absorption_with_materials(ctx):
    """non-uniform densities on both sides: the removed moments are the inertial property updaters' (remove_voxel per emptied voxel)"""
    rng = np.random.default_rng(21)
    objs = []
    for seed, cc, r in ((1, (3, 3, 3), 19.0), (2, (2, 2, 2), 13.0)):
        n = cc[0] * 16
        x, y, z = np.meshgrid(*[np.arange(n) + 0.5] * 3, indexing="ij")
        d = np.sqrt((x - n / 2) ** 2 + (y - n / 2) ** 2 + (z - n / 2) ** 2) - r
        sd = np.clip(np.trunc(d.astype(np.float32) * np.float32(50.0)), -128, 127).astype(np.int8)
        ty = ((np.floor(x / 8) + np.floor(y / 8) * seed).astype(np.int64) % 3).astype(np.uint8)
        ty[sd >= 0] = 255
        sdt, tyt = ol.dense_to_tiled(sd), ol.dense_to_tiled(ty)
        o = ol.OracleObject.from_dense(cc, sdt, tyt)
        g = VoxelObject.from_dense(ctx, cc, sdt, tyt)
        o.update_occupied_voxel_ranges()
        o.compute_all_derived_state()
        g.compute_all_derived_state()
        g.update_occupied_voxel_ranges()
        g.label_regions()
        objs.append((o, g))
    (A, GA), (B, GB) = objs
    da, db = rng.uniform(0.5, 3.0, 256).astype(np.float32), rng.uniform(0.5, 3.0, 256).astype(np.float32)
    qa = np.array([0, 0, 0, 1], np.float32)
    qb = np.array([0.0, 0.0, np.sin(0.25), np.cos(0.25)], np.float32)
    ta = np.array([24.0, 24.0, 24.0], np.float32)
    tb = (np.array([16.0, 16.0, 16.0]) - rot64(qb, np.array([20.0, 14.0, -3.0]))).astype(np.float32)
    ra, rb = absorb_mutual_both(A, GA, qa, ta, B, GB, qb, tb, 1.0, da, db)
    assert ra["emptied_voxels"] > 300 and rb["emptied_voxels"] > 300
    GA.close()
    GB.close()


def test_occupied_ranges_stay_as_the_reference_keeps_them(ctx):
    """the reference refreshes an object's occupied ranges after an edit only when a chunk was removed (intersection.rs:384-386): a bite that
    takes the outermost voxels leaves the ranges wider than needed, and the next edit still visits (and re-quantises) the empty voxels out
    there. Two overlapping bites at the tip of a sphere, then a capsule across it: equal to the oracle each time, and the ranges the HIP path
    reports on an explicit update are the tight ones again"""
    o, g = both(ctx, scenes.sphere_scene(20.0))
    c0 = centre_of(o)
    tip = c0 + np.float32(20.0) * np.array([0.0, 1.0, 0.0], np.float32)
    r1 = absorb_both(o, g, tip, 6.0)
    assert r1["removed_chunks"] == 0 and r1["emptied_by_type"].sum() > 50
    before = o.info()["occupied_voxel_ranges"]
    r2 = absorb_both(o, g, tip - np.array([0.0, 4.0, 0.0], np.float32), 9.0)
    assert r2["removed_chunks"] == 0
    assert o.info()["occupied_voxel_ranges"] == before  # (stale on purpose)
    absorb_capsule_both(o, g, tip + np.array([-15.0, -1.0, 0.0], np.float32), np.array([30.0, 0.0, 0.0], np.float32), 5.0)
    o.update_occupied_voxel_ranges()
    tight = g.update_occupied_voxel_ranges()
    assert [tuple(t) for t in tight] == [tuple(t) for t in o.info()["occupied_voxel_ranges"]]
    assert tight[1][1] < before[1][1]  # the top rows are gone
    g.close()
