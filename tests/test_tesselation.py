"""Rows a13 / f3 (host code in libimpact_voxel_hip.so, no GPU): Delaunay tetrahedralization, Voronoi cells and the fracture-point
sampling. The cases are the reference's own (impact_tesselation/src/delaunay.rs:1826-1970, voronoi.rs:415-690, re-typed as data), the
validity check is tests/tess_check.py (exact rationals)."""
import numpy as np
import pytest

import tess_check as tc
from impact_amd import fracturing as fr


def D(points):
    return fr.DelaunayTetrahedralization(np.asarray(points, dtype=np.float32))


def test_less_than_four_points_is_empty():
    assert D([[0, 0, 0], [1, 0, 0], [0, 1, 0]]).n_tetrahedra == 0
    assert D(np.zeros((0, 3))).n_tetrahedra == 0


def test_coplanar_points_are_empty():
    assert D([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]]).n_tetrahedra == 0


def test_four_points():
    d = D([[-1, 0, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0]])
    assert d.n_tetrahedra == 1
    tc.validate_delaunay(d)


def test_five_points():
    d = D([[-1, 0, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0], [1, 1, 0]])
    assert d.n_tetrahedra == 2
    tc.validate_delaunay(d)


def test_coincident_points_are_ignored():
    d = D([[-1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, 1], [1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 1, 0]])
    assert d.n_tetrahedra == 1
    tc.validate_delaunay(d)


def grid_points(jitter_seed=None):
    rng = np.random.default_rng(jitter_seed) if jitter_seed is not None else None
    pts = []
    for i in range(3):
        for j in range(3):
            for k in range(3):
                p = np.array([i, j, k], dtype=np.float32)
                if rng is not None:
                    p = p + (rng.random(3, dtype=np.float32) - np.float32(0.5))
                pts.append(p)
    return np.array(pts, dtype=np.float32)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_randomized_grid_is_valid(seed):
    d = D(grid_points(seed))
    assert d.n_tetrahedra > 0
    tc.validate_delaunay(d)


def test_regular_grid_is_valid():
    """27 lattice points: every cube's 8 corners are co-spherical, the degenerate case of the in-sphere predicate"""
    d = D(grid_points())
    assert d.n_tetrahedra > 0
    tc.validate_delaunay(d)
    ok, vol, hv = tc.hull_volume_matches(d)
    assert ok, (vol, hv)


@pytest.mark.parametrize("seed,n,scale", [(0, 40, 10.0), (1, 120, 30.0), (2, 300, 100.0), (3, 64, 0.01)])
def test_random_clouds_are_valid(seed, n, scale):
    rng = np.random.default_rng(seed)
    d = D(rng.uniform(-scale, scale, (n, 3)))
    tc.validate_delaunay(d, check_bounding=False)
    assert d.n_vertices == n + 4
    # the tetrahedra fill the hull up to the slivers a finite bounding tetrahedron hides (the reference's construction has the same property)
    ok, vol, hv = tc.hull_volume_matches(d, rtol=2e-2)
    assert ok, (vol, hv)


def test_boundary_face_planes_of_a_cube():
    pts = [[x, y, z] for x in (0, 2) for y in (0, 2) for z in (0, 2)]
    d = D(pts)
    tc.validate_delaunay(d)
    planes = d.compute_boundary_face_planes()
    assert len(planes) == 12
    rows = {tuple(float(v) + 0.0 for v in p) for p in planes}
    for axis in range(3):
        n = [0.0, 0.0, 0.0]
        n[axis] = -1.0
        assert tuple(n + [0.0]) in rows
        n[axis] = 1.0
        assert tuple(n + [2.0]) in rows
    np.testing.assert_array_equal(d.compute_aabb(), [0, 0, 0, 2, 2, 2])


# ---- Voronoi ----------------------------------------------------------------------------------------------------------------------

def test_voronoi_four_points_structure():
    d = D([[-1, 1, 0], [0, 1, 1], [1, 1, 0], [0, 2, 0]])
    for v in d.internal_vertex_indices():
        p = d.voronoi_polyhedron(v)
        assert (len(p["vertices"]), len(p["rays"]), len(p["face_planes"])) == (1, 3, 3)


def test_voronoi_five_points_structure():
    d = D([[-1, 0, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0], [1, 1, 0]])
    single = double = 0
    for v in d.internal_vertex_indices():
        p = d.voronoi_polyhedron(v)
        if len(p["vertices"]) == 1:
            assert (len(p["rays"]), len(p["face_planes"])) == (3, 3)
            single += 1
        else:
            assert len(p["vertices"]) == 2 and len(p["face_planes"]) == 4
            double += 1
    assert (single, double) == (2, 3)


def dedup(rows, eps=1e-5):
    out = []
    for r in rows:
        if not any(np.allclose(r, o, atol=eps, rtol=eps) for o in out):
            out.append(r)
    return out


def test_voronoi_regular_grid_structure():
    pts = grid_points()
    d = D(pts)
    for idx, v in enumerate(d.internal_vertex_indices()):
        i, j, k = idx // 9, (idx // 3) % 3, idx % 3
        on_boundary = sum(1 for x in (i, j, k) if x in (0, 2))
        p = d.voronoi_polyhedron(v)
        verts = dedup(p["vertices"])
        faces = dedup(p["face_planes"])
        dirs = dedup(p["rays"][:, 3:6]) if len(p["rays"]) else []
        expect = {3: (1, 3, (3, 5)), 2: (2, 2, (4, 6)), 1: (4, 1, (5, 7)), 0: (8, 0, (6,))}[on_boundary]
        assert len(verts) == expect[0], (idx, len(verts))
        assert len(dirs) == expect[1]
        assert len(faces) in expect[2]
        axis_aligned = 0
        for f in faces:
            sd = float(np.dot(f[:3], pts[idx]) - f[3])
            if np.isclose(np.abs(f[:3]).max(), 1.0):
                axis_aligned += 1
                assert abs(sd + 0.5) < 1e-5
            else:
                assert abs(sd + 0.5 * np.sqrt(2.0)) < 1e-5
        assert axis_aligned == {3: 3, 2: 4, 1: 5, 0: 6}[on_boundary]


@pytest.mark.parametrize("seed", [0, 1])
def test_voronoi_cells_are_the_nearest_point_regions(seed):
    """every face plane is the perpendicular bisector to a Delaunay neighbour, the generating point is inside all of them, and a random
    probe is inside the cell's planes exactly when that generator is its nearest point"""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-8, 8, (50, 3)).astype(np.float32)
    d = D(pts)
    cells = [d.voronoi_polyhedron(v) for v in d.internal_vertex_indices()]
    P = d.vertices[4:].astype(np.float64)
    for i, c in enumerate(cells):
        for f in c["face_planes"].astype(np.float64):
            assert abs(np.linalg.norm(f[:3]) - 1) < 1e-5
            sd = np.dot(f[:3], P[i]) - f[3]
            assert sd < 0
            mirror = P[i] - 2 * sd * f[:3]  # the neighbour on the other side of the bisector
            assert np.min(np.linalg.norm(P - mirror, axis=1)) < 1e-3 * (1 + abs(sd))
    probes = rng.uniform(-6, 6, (400, 3))
    for q in probes:
        dist = np.linalg.norm(P - q, axis=1)
        order = np.argsort(dist)
        if dist[order[1]] - dist[order[0]] < 1e-3:
            continue
        for i, c in enumerate(cells):
            f = c["face_planes"].astype(np.float64)
            inside = bool(np.all(f[:, :3] @ q - f[:, 3] <= 0))
            assert inside == (i == order[0]), (i, order[0])


BOX = [0, 0, 0, 10, 10, 10]


def poly(vertices, rays=()):
    r = np.array([list(o) + list(np.asarray(dd, dtype=np.float64) / np.linalg.norm(dd)) for o, dd in rays], dtype=np.float32).reshape(-1, 6)
    return {"vertices": np.array(vertices, dtype=np.float32).reshape(-1, 3), "rays": r}


@pytest.mark.parametrize("p,expect", [
    (poly([[5, 5, 5]]), [5, 5, 5, 5, 5, 5]),
    (poly([[15, 5, 5]]), None),
    (poly([[-1, 5, 5], [5, 12, 5], [5, 2, 5], [5, 5, 11]]), [0, 2, 5, 5, 10, 10]),
    (poly([[5, 3, 5]], [([5, 3, 5], [0, 1, 0])]), [5, 3, 5, 5, 10, 5]),
    (poly([[5, -5, 5]], [([5, -5, 5], [0, 1, 0])]), [5, 0, 5, 5, 10, 5]),
    (poly([[8, 15, 2], [15, 3, 8]], [([8, 15, 2], [-1, -0.5, 0]), ([15, 3, 8], [-1, -0.5, 0])]), [0, 0, 2, 10, 10, 8]),
])
def test_bounded_aabb(p, expect):
    got = fr.compute_bounded_aabb(p, BOX)
    if expect is None:
        assert got is None
    else:
        np.testing.assert_allclose(got, expect, atol=1e-5)


# ---- a13: fracture points -----------------------------------------------------------------------------------------------------------

def _impact_case(direction, rotation, magnitude, extent=8.0):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import fracture_points as ofp

    cfg = fr.default_impact_config()
    props = fr.fracturing_properties(1.0e3, 100.0, 0.1, 0.05, 0.5)
    direction = np.asarray(direction, dtype=np.float32)
    direction = direction / np.linalg.norm(direction)
    rotation = np.asarray(rotation, dtype=np.float32)
    rotation = rotation / np.linalg.norm(rotation)
    aabb = np.array([-extent, -extent, -extent, extent, extent, extent], dtype=np.float32)
    args = (4.0, rotation, np.array([0.5, -0.25, 1.0], dtype=np.float32), aabb, np.array([1.0, 2.0, (extent - 0.5) * (-1.0 if direction[2] > 0 else 1.0)], dtype=np.float32), direction, magnitude)
    got = fr.generate_impact_fracture_points(cfg, props, *args, 7)
    c = {k: cfg[0][k] for k in cfg.dtype.names}
    p = {k: props[0][k] for k in props.dtype.names}
    want = ofp.generate(c, p, *args, 7)
    return got, want, cfg


@pytest.mark.parametrize("direction,rotation,magnitude", [
    ([0.0, 0.0, -1.0], [0, 0, 0, 1], 8.0e3),       # antiparallel to +z: the half-turn branch of the arc rotation
    ([0.0, 0.0, 1.0], [0, 0, 0, 1], 6.0e3),        # parallel: identity
    ([0.3, -0.5, -0.8], [0.1, 0.2, -0.3, 0.9], 1.0e4),
])
def test_fracture_points_equal_the_second_restatement(direction, rotation, magnitude):
    (bnd, pts, state), (obnd, opts, ostate), cfg = _impact_case(direction, rotation, magnitude)
    assert len(bnd) == len(obnd) == int(cfg[0]["boundary_polar_grid_size"]) * int(cfg[0]["boundary_azimuthal_grid_size"]) + 1
    assert len(pts) == len(opts) and len(pts) > 4
    assert state == ostate
    np.testing.assert_allclose(bnd, obnd, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(pts, opts, rtol=1e-5, atol=1e-4)


def test_force_below_the_threshold_makes_no_points():
    (bnd, pts, state), (obnd, opts, ostate), _ = _impact_case([0, 0, -1], [0, 0, 0, 1], 900.0)
    assert len(bnd) == len(pts) == len(obnd) == len(opts) == 0 and state == ostate == 7


def test_fracture_points_feed_a_valid_tetrahedralization():
    (bnd, pts, _), _, _ = _impact_case([0.3, -0.5, -0.8], [0.1, 0.2, -0.3, 0.9], 1.0e4)
    d = D(pts)
    tc.validate_delaunay(d, check_bounding=False)
    hull = D(bnd)
    assert hull.n_tetrahedra > 0 and len(hull.compute_boundary_face_planes()) >= 4
    # min distance rule: no two fracture points closer than the smallest allowed fragment extent (normalized)
    dd = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=2) + np.eye(len(pts)) * 1e9
    assert dd.min() > 0
