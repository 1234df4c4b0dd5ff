"""GPU parity of the polyhedron clip (a12: `extract_polyhedron` / `copy_polyhedron`, object/extraction.rs:604-1768)
against the oracle: same outcome, chunk box, origin offset, voxel bytes, chunk kinds, derived state, meshes and
moments for the polyhedron object and (extract mode) for what is left of the parent. Includes BASELINE config 3's
"copy each octant with its 3 cutting planes + 3 far planes"."""
import itertools

import numpy as np
import pytest

import oracle_lib as ol
import parity_util as pu
from impact_amd import scenes
from impact_amd.sdf_graph import SDFGraph, SDFNode
from test_gpu_split import assert_objects_equal, build

pytestmark = pytest.mark.gpu


def box_planes(lo, hi):
    """six axis-aligned face planes of the box [lo, hi] with outward normals"""
    planes = []
    for d in range(3):
        n = [0.0, 0.0, 0.0]
        n[d] = 1.0
        planes.append((*n, float(hi[d])))
        n[d] = -1.0
        planes.append((*n, -float(lo[d])))
    return np.array(planes, dtype=np.float32)


def rotated_box(centre, half, axis, angle):
    """face planes + AABB of an oriented box"""
    axis = np.asarray(axis, dtype=np.float64)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)
    planes = []
    c = np.asarray(centre, dtype=np.float64)
    for d in range(3):
        for sgn in (1.0, -1.0):
            n = (sgn * R[:, d]).astype(np.float32)
            n = n / np.float32(np.linalg.norm(n.astype(np.float64)))
            planes.append((*n, float(np.dot(n.astype(np.float64), c)) + half[d]))
    corners = np.array([c + R @ (np.array(s) * half) for s in itertools.product((-1, 1), repeat=3)])
    return np.array(planes, dtype=np.float32), np.concatenate([corners.min(0), corners.max(0)]).astype(np.float32)


def clip_both(ctx, o, g, planes, aabb, copy, expect=None):
    rc_o, co, org_o = o.clip_polyhedron(planes, aabb, copy=copy)
    rc_g, cg, org_g = (g.copy_polyhedron if copy else g.extract_polyhedron)(aabb, planes)
    assert rc_g == rc_o
    if expect is not None:
        assert rc_o == expect
    if rc_o == 1:
        assert org_g == org_o
        assert_objects_equal(co, cg, "polyhedron: ")
    if not copy and rc_o:
        assert_objects_equal(o, g, "parent: ")
    return rc_o, co, cg


def test_axis_aligned_half_box(ctx):
    """a 40^3 box cut at x = 21: copy, then extract; voxel counts add up"""
    o, g = build(ctx, scenes.box_scene((40.0, 40.0, 40.0)))
    planes = box_planes((-100, -100, -100), (21, 100, 100))
    aabb = (-100, -100, -100, 21, 100, 100)
    _, co, cg = clip_both(ctx, o, g, planes, aabb, copy=True, expect=1)
    cg.close()
    clip_both(ctx, o, g, planes, aabb, copy=False, expect=1)[2].close()
    assert g.count_regions() == 1


def test_oriented_box_through_sphere(ctx):
    """tilted cutting planes through a sphere with two voxel types: boundary voxels get max(sdf, d) / complement"""
    gr = SDFGraph()
    gr.add_node(SDFNode.new_sphere(26.0))
    o, g = build(ctx, gr, extent=0.5)
    planes, aabb = rotated_box((30.0, 24.0, 31.0), np.array([14.0, 9.0, 30.0]), (1.0, 2.0, 0.5), 0.6)
    clip_both(ctx, o, g, planes, aabb, copy=True, expect=1)[2].close()
    rc, co, cg = clip_both(ctx, o, g, planes, aabb, copy=False, expect=1)
    cg.close()
    # a second cut through what is left (chunks converted earlier, regions may have separated)
    planes2, aabb2 = rotated_box((20.0, 30.0, 22.0), np.array([6.0, 25.0, 8.0]), (0.0, 1.0, 1.0), -0.4)
    rc2, co2, cg2 = clip_both(ctx, o, g, planes2, aabb2, copy=False)
    if cg2 is not None:
        cg2.close()


def test_polyhedron_missing_the_object_and_tiny_polyhedra(ctx):
    o, g = build(ctx, scenes.box_scene((30.0, 30.0, 30.0)))
    planes, aabb = rotated_box((200.0, 200.0, 200.0), np.array([5.0, 5.0, 5.0]), (1, 0, 0), 0.3)
    clip_both(ctx, o, g, planes, aabb, copy=False, expect=0)
    # a 1.2 x 1.2 x 0.8 box holding 4 voxel centres: fewer than 8 voxels -> removed from the parent, no object (outcome 2)
    planes, aabb = rotated_box((16.0, 16.0, 16.5), np.array([0.6, 0.6, 0.4]), (0, 0, 1), 0.0)
    clip_both(ctx, o, g, planes, aabb, copy=False, expect=2)
    # 5-voxel cube straddling the chunk corner at (16,16,16): repacked into a single chunk
    planes, aabb = rotated_box((16.0, 16.0, 16.0), np.array([2.5, 2.5, 2.5]), (1, 1, 0), 0.5)
    rc, co, cg = clip_both(ctx, o, g, planes, aabb, copy=True, expect=1)
    assert cg.chunk_counts == (1, 1, 1)
    cg.close()


def test_config3_copy_each_octant(ctx):
    """BASELINE config 3 / SURVEY §8d.3: polyhedron COPY of every octant of the 256^3 fracture body using its three
    cutting planes and three far planes"""
    o, g = build(ctx, scenes.fracture_scene())
    from impact_amd.voxel import SDFVoxelGenerator

    # the cutting planes go through the middle of the 3-voxel slabs, i.e. through the body centre, which sits at
    # shifted_grid_center + 0.5 voxels in normalized model space (voxel centres are at index + 0.5)
    centre = float(SDFVoxelGenerator(1.0, scenes.fracture_scene()).shifted_grid_center[0]) + 0.5
    total = 0
    for sx, sy, sz in itertools.product((-1, 1), repeat=3):
        lo = [centre if s > 0 else -10.0 for s in (sx, sy, sz)]
        hi = [300.0 if s > 0 else centre for s in (sx, sy, sz)]
        planes = box_planes(lo, hi)
        aabb = (*lo, *hi)
        rc, co, cg = clip_both(ctx, o, g, planes, aabb, copy=True, expect=1)
        assert cg.count_regions() == 1
        total += int(np.count_nonzero((co.export_dense()[2] & 1) == 0))
        cg.close()
    assert total == int(np.count_nonzero((o.export_dense()[2] & 1) == 0))


def test_batched_copies_equal_the_looped_ones(ctx):
    """`ivx_copy_polyhedra` (all fragments of an impact in one call) against `copy_polyhedron` one by one and against the oracle: config 3's
    eight octants plus an oriented box, a box that misses the object and a crumb"""
    o, g = build(ctx, scenes.fracture_scene())
    from impact_amd.voxel import SDFVoxelGenerator

    centre = float(SDFVoxelGenerator(1.0, scenes.fracture_scene()).shifted_grid_center[0]) + 0.5
    sets = []
    for sx, sy, sz in itertools.product((-1, 1), repeat=3):
        lo = [centre if s > 0 else -10.0 for s in (sx, sy, sz)]
        hi = [300.0 if s > 0 else centre for s in (sx, sy, sz)]
        sets.append((box_planes(lo, hi), np.array([*lo, *hi], dtype=np.float32)))
    sets.append(rotated_box((90.0, 120.0, 140.0), np.array([30.0, 22.0, 41.0]), (1.0, 2.0, 0.5), 0.6))
    sets.append(rotated_box((900.0, 900.0, 900.0), np.array([5.0, 5.0, 5.0]), (1, 0, 0), 0.3))  # misses
    sets.append(rotated_box((centre + 40.0, centre + 40.0, centre + 40.0), np.array([0.6, 0.6, 0.4]), (0, 0, 1), 0.0))  # 4 voxel centres
    batched = g.copy_polyhedra([s[1] for s in sets], [s[0] for s in sets])
    assert [b[0] for b in batched] == [1] * 9 + [0, 2]
    for (planes, aabb), (rc_b, child_b, org_b) in zip(sets, batched):
        rc_o, co, org_o = o.clip_polyhedron(planes, aabb, copy=True)
        rc_l, child_l, org_l = g.copy_polyhedron(aabb, planes)
        assert rc_b == rc_o == rc_l
        if rc_o == 1:
            assert org_b == org_o == org_l
            assert_objects_equal(co, child_b, "batched child: ")
            child_l.close()
            child_b.close()
    assert_objects_equal(o, g, "parent untouched: ")


@pytest.mark.parametrize("seed", pu.fuzz_seeds([61, 62, 63, 64]))
def test_random_oriented_box_through_random_body(ctx, seed):
    """random SDF bodies (tests/test_gpu_random_sdf.py's trees) cut by a random oriented box somewhere in their grid — missing the body,
    swallowing it, slicing through blends and thin parts: first the copy, then the extraction, polyhedron object and what is left of the
    parent against the oracle (voxel bytes, chunk state, derived state, meshes, moments)"""
    from impact_amd.voxel import SDFVoxelGenerator
    from test_gpu_random_sdf import random_tree

    rng = np.random.default_rng(seed)
    gr = SDFGraph()
    random_tree(gr, rng, int(rng.integers(1, 4)))
    extent = [1.0, 0.5][seed % 2]
    cc = SDFVoxelGenerator(extent, gr, 0).chunk_counts()
    if min(cc) == 0:
        return
    o, g = build(ctx, gr, extent=extent)
    size = np.array(cc, dtype=np.float64) * 16.0 * extent
    centre = rng.uniform(0.1, 0.9, 3) * size
    half = rng.uniform(0.08, 0.6, 3) * size.min()
    planes, aabb = rotated_box(centre, half, rng.normal(size=3), float(rng.uniform(0, 3.1)))
    rc, co, cg = clip_both(ctx, o, g, planes, aabb, copy=True)
    if rc == 1:
        cg.close()
    rc, co, cg = clip_both(ctx, o, g, planes, aabb, copy=False)
    if rc == 1:
        cg.close()
    g.close()
