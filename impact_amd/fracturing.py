"""Host-side mirror of the reference's impact -> fragments chain on top of the C ABI:

  VoxelImpactFracturingConfig, FracturingProperties       impact_voxel/src/interaction/fracturing.rs:61-86, 855-871
  generate_impact_fracture_points                         fracturing.rs:1710-2015
  DelaunayTetrahedralization                              impact_tesselation/src/delaunay.rs
  VoronoiPolyhedron                                       impact_tesselation/src/voronoi.rs
  fracture_voxel_object                                   extract_fracture_region_object + FracturingProcess (fracturing.rs:1047-1240, 1537-1632)

The geometry runs on the host inside libimpact_voxel_hip.so (tesselation.cpp: a few hundred points per impact); the voxels are cut on the
GPU (ivx_clip_polyhedron, ivx_copy_polyhedra)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import check, ptr

NO_TETRAHEDRON = 0xFFFFFFFF


def default_impact_config() -> np.ndarray:
    c = np.zeros(1, dtype=capi.IMPACT_FRACTURING_CONFIG_DTYPE)
    capi.lib().ivx_impact_fracturing_config_default(ptr(c))
    return c


def fracturing_properties(fracturing_force, shattering_pressure, fragment_scale, min_fragment_extent, max_fragment_extent) -> np.ndarray:
    p = np.zeros(1, dtype=capi.FRACTURING_PROPERTIES_DTYPE)
    p[0] = (fracturing_force, shattering_pressure, fragment_scale, min_fragment_extent, max_fragment_extent)
    return p


def generate_impact_fracture_points(config, properties, inverse_voxel_extent, world_to_object_rotation, world_to_object_translation, aabb, force_position,
                                    force_direction, force_magnitude, rng_state: int):
    """-> (boundary points [nb, 3], fracture points [nf, 3], new rng state); normalized (voxel) units"""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    cap_b = int(config[0]["boundary_polar_grid_size"]) * int(config[0]["boundary_azimuthal_grid_size"]) + 1
    cap_f = int(config[0]["max_fragment_count"])
    bnd = np.zeros((cap_b, 3), dtype=np.float32)
    pts = np.zeros((max(cap_f, 1), 3), dtype=np.float32)
    nb, nf = C.c_size_t(0), C.c_size_t(0)
    state = C.c_uint64(rng_state)
    check(capi.lib().ivx_generate_impact_fracture_points(ptr(config), ptr(properties), float(inverse_voxel_extent), ptr(f(world_to_object_rotation)),
                                                         ptr(f(world_to_object_translation)), ptr(f(aabb)), ptr(f(force_position)), ptr(f(force_direction)),
                                                         float(force_magnitude), C.byref(state), ptr(bnd), cap_b, C.byref(nb), ptr(pts), cap_f, C.byref(nf)))
    return bnd[: nb.value].copy(), pts[: nf.value].copy(), int(state.value)


class DelaunayTetrahedralization:
    """`DelaunayTetrahedralization::construct` (delaunay.rs:99-111): vertices 0..3 are the ad-hoc bounding tetrahedron"""

    def __init__(self, points):
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        h = C.c_void_p()
        check(capi.lib().ivx_delaunay_construct(ptr(p) if len(p) else None, len(p), C.byref(h)))
        self.h = h
        cnt = np.zeros(2, dtype=np.uint32)
        check(capi.lib().ivx_delaunay_counts(self.h, ptr(cnt)))
        self.n_vertices, self.n_tetrahedra = int(cnt[0]), int(cnt[1])
        self.vertices = np.zeros((self.n_vertices, 3), dtype=np.float32)
        self.tetrahedra = np.zeros((self.n_tetrahedra, 4), dtype=np.uint32)
        self.neighbors = np.zeros((self.n_tetrahedra, 4), dtype=np.uint32)
        if self.n_vertices:
            check(capi.lib().ivx_delaunay_download(self.h, ptr(self.vertices), ptr(self.tetrahedra) if self.n_tetrahedra else None,
                                                   ptr(self.neighbors) if self.n_tetrahedra else None))

    def displace_vertices(self, offset):
        """`displace_vertices` (fracturing.rs:996-1002): move every vertex by `offset` (f32) after the construction"""
        o = np.ascontiguousarray(offset, dtype=np.float32).reshape(3)
        check(capi.lib().ivx_delaunay_displace_vertices(self.h, ptr(o)))
        if self.n_vertices:
            check(capi.lib().ivx_delaunay_download(self.h, ptr(self.vertices), ptr(self.tetrahedra) if self.n_tetrahedra else None,
                                                   ptr(self.neighbors) if self.n_tetrahedra else None))

    def close(self):
        if getattr(self, "h", None):
            capi.lib().ivx_delaunay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def internal_vertex_indices(self):
        return range(4, self.n_vertices)

    def compute_aabb(self):
        out = np.zeros(6, dtype=np.float32)
        check(capi.lib().ivx_delaunay_aabb(self.h, ptr(out)))
        return out

    def compute_boundary_face_planes(self):
        n = C.c_size_t(0)
        check(capi.lib().ivx_delaunay_boundary_face_planes(self.h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 4), dtype=np.float32)
        check(capi.lib().ivx_delaunay_boundary_face_planes(self.h, ptr(out), n.value, C.byref(n)))
        return out[: n.value]

    def voronoi_polyhedron(self, vertex: int):
        """`VoronoiPolyhedron::extract_from_delaunay_tetrahedra` -> dict(vertices [n, 3], rays [n, 6], face_planes [n, 4])"""
        n = (C.c_size_t * 3)()
        check(capi.lib().ivx_voronoi_polyhedron(self.h, vertex, None, 0, None, 0, None, 0, n))
        v = np.zeros((max(n[0], 1), 3), dtype=np.float32)
        r = np.zeros((max(n[1], 1), 6), dtype=np.float32)
        p = np.zeros((max(n[2], 1), 4), dtype=np.float32)
        check(capi.lib().ivx_voronoi_polyhedron(self.h, vertex, ptr(v), n[0], ptr(r), n[1], ptr(p), n[2], n))
        return {"vertices": v[: n[0]], "rays": r[: n[1]], "face_planes": p[: n[2]]}


def compute_bounded_aabb(polyhedron, bounding_aabb):
    """`VoronoiPolyhedron::compute_bounded_aabb` (voronoi.rs:254-318) -> 6 floats or None"""
    v, r = np.ascontiguousarray(polyhedron["vertices"]), np.ascontiguousarray(polyhedron["rays"])
    out = np.zeros(6, dtype=np.float32)
    has = C.c_int(0)
    bb = np.ascontiguousarray(bounding_aabb, dtype=np.float32)
    check(capi.lib().ivx_voronoi_bounded_aabb(ptr(v) if len(v) else None, len(v), ptr(r) if len(r) else None, len(r), ptr(bb), ptr(out), C.byref(has)))
    return out if has.value else None


def fragment_plane_sets(fracture_points, bounding_aabb, shift: float = -0.1, displacement=None):
    """Delaunay of the fracture points -> per point its Voronoi cell as (planes displaced by `shift`, bounded box), cells without
    overlap left out (FracturingProcess::initialize + generate_fragment, fracturing.rs:976-994, 1190-1207). `displacement`: moved by that much
    AFTER the construction (offset_tetrahedralization_to_fracture_region_object, fracturing.rs:996-1002: the reference builds on the points as
    sampled and then displaces the vertices). -> (list of (vertex, planes, aabb), tetrahedralization)"""
    d = DelaunayTetrahedralization(fracture_points)
    if displacement is not None:
        d.displace_vertices(displacement)
    out = []
    for v in d.internal_vertex_indices():
        poly = d.voronoi_polyhedron(v)
        bb = compute_bounded_aabb(poly, bounding_aabb)
        if bb is None or len(poly["face_planes"]) == 0:
            continue
        planes = poly["face_planes"].copy()
        planes[:, 3] = planes[:, 3] + np.float32(shift)  # displace_along_normal
        out.append((v, planes, bb))
    return out, d


def fracture_voxel_object(obj, boundary_points, fracture_points):
    """The voxel side of one impact: the fracture region (hull of the boundary points) is EXTRACTED from the object, the Voronoi cells of the
    fracture points (shrunk by 0.1 voxel) are COPIED out of the region object in one batched call, the region object is dropped.
    -> dict(region_outcome, region_origin, fragments [(child VoxelObject, origin offset in the ORIGINAL object, fracture point index)])"""
    region_tets = DelaunayTetrahedralization(boundary_points)
    if region_tets.n_tetrahedra == 0:
        return {"region_outcome": 0, "fragments": []}
    cc = np.asarray(obj.chunk_counts, dtype=np.float32) * 16.0
    object_aabb = np.array([0.0, 0.0, 0.0, cc[0], cc[1], cc[2]], dtype=np.float32)  # compute_normalized_chunk_grid_bounds
    ra = region_tets.compute_aabb()
    lo, hi = np.maximum(ra[:3], object_aabb[:3]), np.minimum(ra[3:], object_aabb[3:])
    if np.any(hi - lo < 0):  # AxisAlignedBox::compute_overlap_with: None only for a negative extent
        return {"region_outcome": 0, "fragments": []}
    outcome, region, origin = obj.extract_polyhedron(np.concatenate([lo, hi]), region_tets.compute_boundary_face_planes())
    if outcome != 1:
        return {"region_outcome": outcome, "fragments": []}
    # the fracture points in the region object's frame (offset_tetrahedralization_to_fracture_region_object, fracturing.rs:996-1002)
    # (built on the points as given, THEN displaced: the bounding sphere, the minimum point separation and the predicates see the original coordinates)
    rcc = np.asarray(region.chunk_counts, dtype=np.float32) * 16.0
    sets, tets = fragment_plane_sets(np.asarray(fracture_points, dtype=np.float32), np.array([0.0, 0.0, 0.0, rcc[0], rcc[1], rcc[2]], dtype=np.float32),
                                     displacement=-np.asarray(origin, dtype=np.float32))
    frags = []
    if sets:
        res = region.copy_polyhedra([s[2] for s in sets], [s[1] for s in sets])
        for (v, _, _), (rc, child, off) in zip(sets, res):
            if rc == 1:
                frags.append((child, tuple(int(a + b) for a, b in zip(off, origin)), v - 4))
    region.close()
    tets.close()
    region_tets.close()
    return {"region_outcome": 1, "region_origin": origin, "fragments": frags, "plane_sets": sets}
