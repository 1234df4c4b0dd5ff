"""GPU parity of the impact -> fragments chain (rows a13 + f3 on top of a12): fracture points -> Delaunay -> Voronoi cells -> polyhedron
extraction / batched copies on the GPU, against the oracle's `clip_polyhedron` with the SAME plane sets. Holds the reference's
`fuzz_failure` inputs (object/extraction.rs:2465-2582: a capsule at voxel extent 7.83997 and 23 points, several of them nearly
coincident or far outside the object)."""
import numpy as np
import pytest

import oracle_lib as ol
import parity_util as pu
from impact_amd import fracturing as fr
from impact_amd import scenes
from impact_amd.sdf_graph import SDFGraph, SDFNode
from test_gpu_clip import clip_both
from test_gpu_split import assert_objects_equal, build

pytestmark = pytest.mark.gpu

FUZZ_POINTS = np.array([
    [9.260002, -100.0, -94.9], [5.4800034, -100.0, 53.58], [10.580002, 10.580002, 8.040001], [10.580002, 6.8600082, -18.099998],
    [-41.2, 10.580002, -40.62], [6.640007, 10.580002, 10.580002], [10.580002, 5.459999, 10.580002], [10.580002, -18.099998, 10.580002],
    [10.380005, 10.539993, 10.580002], [53.559998, 10.580002, 10.380005], [10.580002, 10.580002, 2.0400085], [10.580002, 10.580002, 10.559998],
    [10.580002, 6.640007, 10.580002], [10.580002, 10.580002, 5.459999], [10.580002, 5.7800064, 6.640007], [10.580002, -59.4, 10.580002],
    [10.580002, 10.580002, 10.580002], [10.559998, 10.580002, 10.580002], [5.7800064, 10.580002, 10.580002], [7.920006, 10.580002, 10.580002],
    [-45.74, 5.4999924, -94.9], [10.580002, 6.760002, 10.580002], [10.580002, -100.0, -100.0]], dtype=np.float32)


def grid_bounds(obj_chunk_counts):
    cc = np.asarray(obj_chunk_counts, dtype=np.float32) * 16.0
    return np.array([0, 0, 0, cc[0], cc[1], cc[2]], dtype=np.float32)


def test_reference_fuzz_failure_case(ctx):
    """every Voronoi cell of the 23 points is EXTRACTED in turn (the reference's loop); after each: polyhedron object and what is left of
    the parent equal the oracle's, and the moments that left the parent are the cell's moments (1e-2 like the reference's own check)"""
    g = SDFGraph()
    g.add_node(SDFNode.new_capsule(7.695157, 11.359792))
    o, gobj = build(ctx, g, extent=7.83997)
    dens = np.ones(256, dtype=np.float32)
    _, m_start = o.inertia(dens)
    tets = fr.DelaunayTetrahedralization(FUZZ_POINTS)
    assert tets.n_tetrahedra > 0
    bounds = grid_bounds(o.chunk_counts)
    extracted = 0
    mass_moved = 0.0
    for v in tets.internal_vertex_indices():
        poly = tets.voronoi_polyhedron(v)
        bb = fr.compute_bounded_aabb(poly, bounds)
        if bb is None:
            continue
        rc, co, cg = clip_both(ctx, o, gobj, poly["face_planes"], bb, copy=False)
        if rc == 1:
            extracted += 1
            _, mc = co.inertia(dens)
            mass_moved += float(mc[0])
            cg.close()
    assert extracted >= 1
    _, m_end = o.inertia(dens)
    assert abs((float(m_end[0]) + mass_moved) - float(m_start[0])) <= 1e-2 * float(m_start[0])


def sphere_object(ctx, radius=40.0):
    g = SDFGraph()
    g.add_node(SDFNode.new_sphere(radius))
    return build(ctx, g)


def impact_points(o, magnitude, seed=3):
    """an impact on top of the sphere, straight down (object frame = world frame shifted by the grid centre)"""
    cc = np.asarray(o.chunk_counts, dtype=np.float32) * 16.0
    centre = cc * 0.5
    cfg = fr.default_impact_config()
    props = fr.fracturing_properties(1.0e3, 10.0, 0.03, 0.2, 0.5)
    aabb = np.array([0, 0, 0, cc[0], cc[1], cc[2]], dtype=np.float32)
    pos = np.array([centre[0] + 3.0, centre[1] - 2.0, centre[2] + 39.0], dtype=np.float32)
    return fr.generate_impact_fracture_points(cfg, props, 1.0, [0, 0, 0, 1], [0, 0, 0], aabb, pos, [0.0, 0.0, -1.0], magnitude, seed)


def test_impact_on_a_sphere_end_to_end(ctx):
    """fracture_voxel_object (extract the region, batched copies of the shrunk cells) against the oracle running the same plane sets one
    by one: region object, remaining parent and every fragment are equal; the fragments are disjoint pieces of the region"""
    o, g = sphere_object(ctx)
    bnd, pts, _ = impact_points(o, 6.0e3)
    assert len(pts) >= 8
    # oracle side: region extraction, then copies of each shrunk cell out of the region object
    region_tets = fr.DelaunayTetrahedralization(bnd)
    ra = region_tets.compute_aabb()
    bounds = grid_bounds(o.chunk_counts)
    raabb = np.concatenate([np.maximum(ra[:3], bounds[:3]), np.minimum(ra[3:], bounds[3:])])
    rc_o, region_o, origin_o = o.clip_polyhedron(region_tets.compute_boundary_face_planes(), raabb, copy=False)
    assert rc_o == 1
    res = fr.fracture_voxel_object(g, bnd, pts)
    assert res["region_outcome"] == 1 and tuple(res["region_origin"]) == tuple(origin_o)
    assert_objects_equal(o, g, "parent after the region left: ")
    sets = res["plane_sets"]
    got = {idx: (child, off) for child, off, idx in res["fragments"]}
    n_frag = 0
    voxels = 0
    for v, planes, bb in sets:
        rc, co, org = region_o.clip_polyhedron(planes, bb, copy=True)
        assert (rc == 1) == ((v - 4) in got)
        if rc == 1:
            child, off = got[v - 4]
            assert tuple(off) == tuple(int(a + b) for a, b in zip(org, origin_o))
            assert_objects_equal(co, child, f"fragment of point {v - 4}: ")
            voxels += int(np.count_nonzero((co.export_dense()[2] & 1) == 0))
            n_frag += 1
            child.close()
    assert n_frag >= 4
    # shrunk cells do not overlap: together they hold fewer voxels than the region
    assert voxels <= int(np.count_nonzero((region_o.export_dense()[2] & 1) == 0))


def test_weak_impact_leaves_the_object_alone(ctx):
    o, g = sphere_object(ctx, 20.0)
    bnd, pts, _ = impact_points(o, 500.0)
    assert len(bnd) == 0 and len(pts) == 0
    res = fr.fracture_voxel_object(g, bnd, pts)
    assert res["region_outcome"] == 0 and res["fragments"] == []
    assert_objects_equal(o, g, "untouched: ")
