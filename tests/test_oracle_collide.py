"""The oracle's restatement of contact generation between two voxel objects (oracle/src/orc_collide.cpp) against independent
checks: the collision probes against a pure-Python f32 walk over the mesh (the reference has no golden vectors for them), the
mutual contacts against a brute-force f64 evaluation of the trilinear SDF at every probe."""
import numpy as np
import pytest

import oracle_lib as ol
from impact_amd import scenes

f32 = np.float32


def make(graph, extent=1.0):
    o = ol.OracleObject.from_sdf(graph, extent, 0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    o.extent = float(extent)
    return o


def probes_python(o, mesh):
    """add_points_for_vertices_in_blocks (collidable.rs:614-731) per submesh, in plain Python with f32 scalars"""
    occ = o.info()["occupied_voxel_ranges"]
    me = min(b - a for a, b in occ)
    log2_bs = 3 if me >= 16 else 2 if me >= 8 else 1 if me >= 4 else 0
    lcb = 4 - log2_bs
    inv = f32(1.0) / f32(o.extent)
    pts, ents = [], []
    P, N, I = mesh.positions, mesh.normals, mesh.indices

    def dot(a, b):
        return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))

    for sm in mesh.submeshes:
        ioff, icnt, voff, vcnt = int(sm[3]), int(sm[4]), int(sm[13]), int(sm[14])
        s = [f32(0.0)] * vcnt
        c = [f32(0.0)] * vcnt
        for t in range(ioff, ioff + icnt, 3):
            i0, i1, i2 = (int(I[t + q]) - voff for q in range(3))
            v0, v1, v2 = P[voff + i0], P[voff + i1], P[voff + i2]
            n0, n1, n2 = N[voff + i0], N[voff + i1], N[voff + i2]
            e01, e12, e20 = v1 - v0, v2 - v1, v0 - v2
            s[i0] = f32(s[i0] + f32(dot(n0, e01) - dot(n0, e20)))
            s[i1] = f32(s[i1] + f32(dot(n1, e12) - dot(n1, e01)))
            s[i2] = f32(s[i2] + f32(dot(n2, e20) - dot(n2, e12)))
            for q in (i0, i1, i2):
                c[q] = f32(c[q] + f32(2.0))
        best = {}
        lo = [f32(int(sm[d]) * 16) for d in range(3)]
        hi = [f32((int(sm[d]) + 1) * 16) for d in range(3)]
        for v in range(vcnt):
            if c[v] == 0:
                continue
            p = P[voff + v]
            vi = [int(min(max(f32(p[d] * inv), lo[d]), hi[d])) for d in range(3)]
            b = [(x & 15) >> log2_bs for x in vi]
            blk = (b[0] << (2 * lcb)) + (b[1] << lcb) + b[2]
            cur = f32(s[v] / c[v])
            if blk not in best or cur < best[blk][1]:
                best[blk] = (p.copy(), cur)
        if best:
            start = len(pts)
            for blk in sorted(best):
                pts.append(best[blk][0])
            ents.append((int(sm[0]), int(sm[1]), int(sm[2]), start, len(pts)))
    return np.array(pts, dtype=f32).reshape(-1, 3), np.array(ents, dtype=np.uint32).reshape(-1, 5)


@pytest.mark.parametrize("case", ["sphere_block8", "small_block4", "thin_block1", "half_extent"])
def test_collision_probes_match_python_walk(case):
    if case == "sphere_block8":
        o = make(scenes.sphere_scene(14.0))
    elif case == "small_block4":
        o = make(scenes.box_scene((9.0, 12.0, 20.0)))
    elif case == "thin_block1":
        o = make(scenes.box_scene((3.0, 20.0, 18.0)))
    else:
        o = make(scenes.sphere_scene(11.0), 0.5)
    mesh = o.mesh()
    pts, ents = o.collision_probes(mesh)
    want_pts, want_ents = probes_python(o, mesh)
    np.testing.assert_array_equal(ents, want_ents)
    np.testing.assert_array_equal(pts.view(np.uint32), want_pts.view(np.uint32))
    assert len(pts) > 8
    # every probe is a mesh vertex of its chunk, at most one per block
    occ = o.info()["occupied_voxel_ranges"]
    me = min(b - a for a, b in occ)
    n_blocks = {3: 8, 2: 64, 1: 512, 0: 4096}[3 if me >= 16 else 2 if me >= 8 else 1 if me >= 4 else 0]
    vset = {tuple(v.view(np.uint32)) for v in mesh.positions}
    for e in ents:
        assert 0 < e[4] - e[3] <= n_blocks
    assert all(tuple(p.view(np.uint32)) in vset for p in pts)


def rot64(q, v):
    x, y, z, w = [float(a) for a in q]
    b = np.array([x, y, z])
    return v * (w * w - b @ b) + b * (2 * (v @ b)) + np.cross(b, v) * (2 * w)


def dense_sd(o):
    sdf, typ, flg, _, info = o.export_dense()
    cc = o.chunk_counts
    sd = ol.tiled_to_dense(sdf, cc).astype(np.float64) * 0.02
    kind = np.repeat(np.repeat(np.repeat(info["kind"].reshape(cc), 16, 0), 16, 1), 16, 2)
    sd[kind == 0] = 2.54  # void chunks read as maximally outside, uniform ones as maximally inside
    sd[kind == 1] = -2.56
    return sd, kind


def trilinear(sd, p):
    lp = p - 0.5
    l = np.floor(lp).astype(int)
    if (l < 0).any() or (l + 1 >= np.array(sd.shape)).any():
        return None
    o = lp - l
    c = sd[l[0]:l[0] + 2, l[1]:l[1] + 2, l[2]:l[2] + 2]
    cx = c[0] * (1 - o[0]) + c[1] * o[0]
    cy = cx[0] * (1 - o[1]) + cx[1] * o[1]
    return cy[0] * (1 - o[2]) + cy[1] * o[2]


def brute_mutual(A, pa, qa, ta, B, qb, tb, which):
    """probes of the probing object whose world position lies inside the other object's trilinear SDF (f64)"""
    ext_p, ext_s = A.extent, B.extent
    sd, kind = dense_sd(B)
    qai = np.array([-qa[0], -qa[1], -qa[2], qa[3]], dtype=np.float64)
    out = {}
    for p in pa[0].astype(np.float64):
        w = rot64(qai, p - ta.astype(np.float64))
        ps = (rot64(qb, w) + tb.astype(np.float64)) / ext_s
        v = trilinear(sd, ps)
        if v is None:
            continue
        c = np.floor(ps).astype(int)
        if kind[tuple(c)] == 0:
            continue
        key = (which, *[int(x) for x in np.floor(p / ext_p)])
        out[key] = (v, w)
    return out


@pytest.mark.parametrize("case", ["two_spheres", "sphere_into_box_rotated", "mixed_extents", "apart"])
def test_mutual_contacts_match_brute_force_sdf_probe(case):
    if case == "two_spheres":
        A, B = make(scenes.sphere_scene(14.0)), make(scenes.sphere_scene(10.0))
        sep, ang = 19.5, 0.0
    elif case == "sphere_into_box_rotated":
        A, B = make(scenes.box_scene((40.0, 12.0, 40.0))), make(scenes.sphere_scene(12.0))
        sep, ang = 13.0, 0.6
    elif case == "mixed_extents":
        A, B = make(scenes.sphere_scene(14.0), 0.5), make(scenes.sphere_scene(9.0), 1.0)
        sep, ang = 11.0, -0.4
    else:
        A, B = make(scenes.sphere_scene(10.0)), make(scenes.sphere_scene(10.0))
        sep, ang = 40.0, 0.3
    ma, mb = A.mesh(), B.mesh()
    pa, pb = A.collision_probes(ma), B.collision_probes(mb)
    ca, cb = A.center_of_mass(), B.center_of_mass()
    # world -> object transforms: A sits with its centre of mass at the world origin, B `sep` along +y and rotated
    axis = np.array([0.3, 0.1, 1.0]) / np.linalg.norm([0.3, 0.1, 1.0])
    qa = np.array([0, 0, 0, 1], dtype=f32)
    ta = ca.copy()
    qb = np.array([*(axis * np.sin(ang / 2)), np.cos(ang / 2)], dtype=f32)
    tb = (cb.astype(np.float64) - rot64(qb, np.array([0.0, sep, 0.0]))).astype(f32)
    wi, pos, nrm, dep = A.mutual_contacts(pa, ca, qa, ta, B, pb, cb, qb, tb)
    if case == "apart":
        assert len(wi) == 0
        return
        assert len(wi) > 10 and set(wi[:, 0].tolist()) == {0, 1}
    want = brute_mutual(A, pa, qa, ta, B, qb, tb, 0)
    want.update(brute_mutual(B, pb, qb, tb, A, qa, ta, 1))
    got = {tuple(int(x) for x in r): (d, p, n) for r, d, p, n in zip(wi, dep, pos, nrm)}
    ext = {0: B.extent, 1: A.extent}
    # every probe clearly inside the other object yields a contact; every contact is (within rounding) inside
    for key, (v, w) in want.items():
        if v < -1e-4:
            assert key in got, key
    for key, (d, p, n) in got.items():
        assert key in want
        v, w = want[key]
        assert v < 1e-4
        np.testing.assert_allclose(p, w, atol=1e-4)
        if v > -2.5:  # (deeper: the capped distance and the centre-of-mass direction are used instead)
            np.testing.assert_allclose(d, -v * ext[key[0]], atol=2e-4)
        assert abs(np.linalg.norm(n) - 1.0) < 1e-5
    # the normals point from B towards A: for A's probes the outward normal of B, for B's probes minus the outward normal of A
    # (for spheres: the radial direction at the probe)
    if case == "two_spheres":
        centre_b_world = rot64(np.array([-qb[0], -qb[1], -qb[2], qb[3]], dtype=np.float64), cb.astype(np.float64) - tb.astype(np.float64))
        for key, (d, p, n) in got.items():
            if d > 1.5:  # (corner samples clamp at -2.56 voxels; the gradient of a partly clamped cell is not radial)
                continue
            radial = (p - centre_b_world) if key[0] == 0 else -(p.astype(np.float64))  # (A's centre of mass is the world origin)
            assert n @ (radial / np.linalg.norm(radial)) > 0.9
    # order: A's probes first, in probe order
    first_b = int(np.argmax(wi[:, 0] == 1))
    assert (wi[:first_b, 0] == 0).all() and (wi[first_b:, 0] == 1).all()
