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
