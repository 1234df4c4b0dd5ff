"""Synthetic, deterministic SDF scenes for BASELINE.json's configs.

No noise nodes (simdnoise parity is unpinned), single voxel type 0, voxel extent 1.0.
"""
from __future__ import annotations

import itertools

from .sdf_graph import SDFGraph, SDFNode


def box_scene(extents=(30.0, 30.0, 30.0)) -> SDFGraph:
    """Config 1 (plumbing): `SDFNode::Box([30,30,30])` -> 32^3 grid, 8 chunks."""
    g = SDFGraph()
    g.add_node(SDFNode.new_box(extents))
    return g


def sphere_scene(radius=100.0) -> SDFGraph:
    """The reference's own benchmark body (engine/src/benchmark/benchmarks/voxel_object.rs:45-131)."""
    g = SDFGraph()
    g.add_node(SDFNode.new_sphere(radius))
    return g


def two_spheres_scene(radius=25.0, separation=60.0) -> SDFGraph:
    """Two disjoint spheres (extraction.rs:2587-2624): must split into two regions."""
    g = SDFGraph()
    s1 = g.add_node(SDFNode.new_sphere(radius))
    s2 = g.add_node(SDFNode.new_sphere(radius))
    t2 = g.add_node(SDFNode.new_translation(s2, (separation, 0.0, 0.0)))
    g.add_node(SDFNode.new_union(s1, t2, 0.0))
    return g


def _asteroid_body(g: SDFGraph, scale: float) -> int:
    """Config-2 body before the optional `Scaling`: smooth union of a core sphere and six bumps,
    minus the (hard) union of eight crater spheres, smoothness 4."""
    core = g.add_node(SDFNode.new_sphere(96.0))
    acc = core
    for axis in range(3):
        for sign in (1.0, -1.0):
            t = [0.0, 0.0, 0.0]
            t[axis] = sign * 80.0
            s = g.add_node(SDFNode.new_sphere(40.0))
            ts = g.add_node(SDFNode.new_translation(s, t))
            acc = g.add_node(SDFNode.new_union(acc, ts, 8.0))
    craters = None
    for sx, sy, sz in itertools.product((1.0, -1.0), repeat=3):
        s = g.add_node(SDFNode.new_sphere(18.0))
        ts = g.add_node(SDFNode.new_translation(s, (55.0 * sx, 55.0 * sy, 55.0 * sz)))
        craters = ts if craters is None else g.add_node(SDFNode.new_union(craters, ts, 0.0))
    body = g.add_node(SDFNode.new_subtraction(acc, craters, 4.0))
    if scale != 1.0:
        body = g.add_node(SDFNode.new_scaling(body, scale))
    return body


def asteroid_scene(scale: float = 1.0) -> SDFGraph:
    """Config 2 (scale 1 -> 256^3 stored grid), headline 512^3 (scale 2), config 5 (scale 4 -> 1024^3)."""
    g = SDFGraph()
    _asteroid_body(g, scale)
    return g


def fracture_scene(scale: float = 1.0) -> SDFGraph:
    """Config 3: the config-2 body cut by three 3-voxel slabs -> exactly 8 disconnected octants."""
    g = SDFGraph()
    body = _asteroid_body(g, 1.0)
    for ext in ((3.0, 260.0, 260.0), (260.0, 3.0, 260.0), (260.0, 260.0, 3.0)):
        slab = g.add_node(SDFNode.new_box(ext))
        body = g.add_node(SDFNode.new_subtraction(body, slab, 0.0))
    if scale != 1.0:
        g.add_node(SDFNode.new_scaling(body, scale))
    return g


def asteroid_row_scene(n: int, scale: float = 2.05, bar_width: float | None = None) -> SDFGraph:
    """Weak-scaling workload: `n` config-2 asteroids (each scaled by `scale`) in a row along x, one per
    x-slab, joined by a thin box so that the body is ONE connected region that crosses every slab
    boundary. The pitch is the single asteroid's own stored grid width (a whole number of chunks), so
    with n ranks every slab holds exactly the N=1 workload: global grid = (n * pitch) x pitch x pitch."""
    from .voxel import SDFVoxelGenerator  # host-side graph compile only

    pitch = 16.0 * SDFVoxelGenerator(1.0, asteroid_scene(scale)).chunk_counts()[0]
    g = SDFGraph()
    acc = None
    for r in range(n):
        body = _asteroid_body(g, scale)
        if n > 1:
            body = g.add_node(SDFNode.new_translation(body, (pitch * (r - 0.5 * (n - 1)), 0.0, 0.0)))
        acc = body if acc is None else g.add_node(SDFNode.new_union(acc, body, 0.0))
    if n > 1:
        w = 12.0 * scale if bar_width is None else bar_width
        bar = g.add_node(SDFNode.new_box((pitch * (n - 1), w, w)))
        g.add_node(SDFNode.new_union(acc, bar, 0.0))
    return g


# ---- rigid-body pile (BASELINE config 4) ----------------------------------------------------------------
def _splitmix(state: int) -> int:
    """impact_math/src/random/splitmix.rs:4-10"""
    m = (1 << 64) - 1
    state = (state + 0x9E3779B97F4A7C15) & m
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def contact_id(a: int, b: int, *indices: int) -> int:
    """`ContactID::from_two_u64_and_n_indices` (impact_physics/src/constraint/contact.rs:180-199)"""
    cid = _splitmix(a ^ _splitmix(b))
    for i in indices:
        cid = _splitmix(cid ^ _splitmix(i))
    return cid


def sphere_pile_scene(n: int = 16, spacing: float = 0.95, radius: float = 0.5, points_per_pair: int = 4, gravity: float = -9.81,
                      restitution: float = 0.4, static_friction: float = 0.7, dynamic_friction: float = 0.5):
    """Config 4: an n^3 lattice of unit-density spheres (5 % overlap at spacing 0.95, cf. the reference's
    `setup_stationary_overlapping_spheres`, engine/src/benchmark/benchmarks/constraint.rs:308-329) with the
    6-neighbour contacts as an explicit list: per touching pair `points_per_pair` contact points on a square
    inside the overlap circle, in the order (body_a, body_b, k); sphere-sphere geometry as in
    impact_physics/src/collision/collidable/sphere.rs:105-136 (normal from B to A, position on B).
    n = 16 -> 4096 bodies, 11 520 pairs, 46 080 contacts. Returns (bodies, contacts)."""
    import numpy as np

    from .capi import CONTACT_DTYPE, CONTACT_MANIFOLD_START
    from .physics import uniform_sphere_body

    f32 = np.float32
    idx = np.arange(n ** 3)
    ix, iy, iz = idx // (n * n), (idx // n) % n, idx % n
    pos = np.stack([ix, iy, iz], axis=1).astype(np.float32) * f32(spacing)
    proto = uniform_sphere_body(radius, 1.0, (0.0, 0.0, 0.0))
    bodies = np.repeat(np.array([proto]), n ** 3)
    bodies["position"] = pos
    bodies["total_force"][:, 1] = f32(gravity) * proto["mass"]
    pairs = []
    for axis, stride in ((2, 1), (1, n), (0, n * n)):  # +z, +y, +x neighbours of body a, sorted below
        coord = (ix, iy, iz)[axis]
        a = idx[coord < n - 1]
        pairs.append(np.stack([a, a + stride, np.full_like(a, axis)], axis=1))
    pairs = np.concatenate(pairs)
    pairs = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
    depth = f32(2.0 * radius) - f32(spacing)
    rho = f32(0.1)
    offs = [(-1, -1), (1, -1), (1, 1), (-1, 1)][:points_per_pair] if points_per_pair <= 4 else None
    contacts = np.zeros(len(pairs) * points_per_pair, dtype=CONTACT_DTYPE)
    k = 0
    for a, b, axis in pairs:
        normal = np.zeros(3, dtype=np.float32)
        normal[axis] = -1.0  # (centre_a - centre_b) / distance: b = a + stride lies at larger coordinate
        u = np.zeros(3, dtype=np.float32)
        v = np.zeros(3, dtype=np.float32)
        u[(axis + 1) % 3] = 1.0
        v[(axis + 2) % 3] = 1.0
        base = pos[b] + f32(radius) * normal
        for q, (su, sv) in enumerate(offs):
            c = contacts[k]
            c["id"] = contact_id(int(a), int(b), q)
            c["body_a"], c["body_b"] = a, b
            c["position"] = base + rho * (f32(su) * u + f32(sv) * v)
            c["normal"] = normal
            c["depth"] = depth
            c["restitution"], c["static_friction"], c["dynamic_friction"] = restitution, static_friction, dynamic_friction
            c["flags"] = CONTACT_MANIFOLD_START if q == 0 else 0
            k += 1
    return bodies, contacts


def pile_churn_frames(contacts, frames: int, points_per_pair: int = 4, fraction: float = 0.1, seed: int = 3):
    """A contact set that changes every frame, as a moving pile's does (the reference prepares whatever contacts the frame produced:
    constraint/solver.rs:386-452): frame f's list is `contacts` (manifolds of `points_per_pair` points, the generator's order) without a random
    `fraction` of its manifolds — so the tenth that was missing the frame before is back, in its place in the generator's order, and another
    tenth is gone. Returns the list of the frames' contact arrays."""
    import numpy as np

    man = contacts.reshape(-1, points_per_pair)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(frames):
        gone = rng.choice(len(man), int(fraction * len(man)), replace=False)
        keep = np.ones(len(man), dtype=bool)
        keep[gone] = False
        out.append(np.ascontiguousarray(man[keep].reshape(-1)))
    return out


def plates_scene(n_chunks: int = 32, thickness: float = 6.0, holes: bool = True) -> SDFGraph:
    """All-surface workload: `n_chunks` parallel plates of `thickness` voxels, one per chunk layer along z, each spanning the
    whole grid in x and y, perforated by a few capsule-shaped holes through the stack — every chunk of the (16 n)^3 stored grid
    holds part of a plate's surface, so every chunk is NonUniform and meshed and the per-voxel byte accounting of SURVEY §8d
    applies to the whole grid (the config-2 asteroid is a solid body: 85 % of its chunks are 8-byte records)."""
    g = SDFGraph()
    side = 16.0 * n_chunks - 3.0  # + 2 border voxels, rounded up -> exactly n_chunks chunks per axis
    acc = None
    for p in range(n_chunks):
        z = 16.0 * (p - 0.5 * (n_chunks - 1))
        plate = g.add_node(SDFNode.new_translation(g.add_node(SDFNode.new_box((side, side, thickness))), (0.0, 0.0, z)))
        acc = plate if acc is None else g.add_node(SDFNode.new_union(acc, plate, 0.0))
    if holes:
        span = 16.0 * n_chunks
        for q, (x, y, r) in enumerate(((0.21, 0.17, 0.055), (-0.27, 0.08, 0.04), (0.05, -0.3, 0.07), (-0.16, -0.22, 0.03))):
            c = g.add_node(SDFNode.new_capsule(span, r * span))
            c = g.add_node(SDFNode.new_rotation_from_axis_angle(c, (1.0, 0.0, 0.0), 1.5707963267948966))  # capsule axis y -> z
            c = g.add_node(SDFNode.new_translation(c, (x * span, y * span, 0.0)))
            acc = g.add_node(SDFNode.new_subtraction(acc, c, 0.0 if q % 2 else 2.0))
    return g
