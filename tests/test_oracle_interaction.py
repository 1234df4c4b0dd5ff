"""The oracle's restatement of the rigid-body bookkeeping after voxel removal (interaction.rs:405-602) against
conservation laws evaluated independently in f64: splitting an object into parts partitions mass, momentum and angular momentum."""
import numpy as np

import oracle_lib as ol
from impact_amd import scenes

f32 = np.float32


def rotm(q):
    x, y, z, w = [float(a) for a in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def state(body):
    """mass, position, velocity, world inertia, angular velocity, angular momentum of a body record in f64"""
    m = float(body["mass"])
    R = rotm(body["orientation"])
    I = body["inertia"].astype(np.float64).reshape(3, 3).T
    Iw = R @ I @ R.T
    L = body["angular_momentum"].astype(np.float64)
    return m, body["position"].astype(np.float64), body["momentum"].astype(np.float64) / m, Iw, np.linalg.solve(Iw, L), L


def test_offset_reference_point_matches_moments_about_the_shifted_origin():
    """offset_reference_point_by (object/inertia.rs:257-267): moments of a voxel body about (0,0,0) moved to another point equal the
    moments computed about that point directly (numpy over the voxels, f64)"""
    o = ol.OracleObject.from_sdf(scenes.box_scene((20.0, 14.0, 18.0)), 0.5, 0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    m32 = o.inertia()[0]
    off = np.array([3.5, -2.0, 6.25], dtype=f32)
    got = ol.offset_reference_point(m32, off)
    sdf, typ, flg, _, info = o.export_dense()
    solid = (ol.tiled_to_dense(flg, o.chunk_counts) & 1) == 0
    e = 0.5
    idx = np.argwhere(solid).astype(np.float64)
    lo, hi = idx * e - off, (idx + 1) * e - off  # voxel cubes relative to the new origin
    vol = e ** 3
    c1 = (lo + hi) / 2  # mean of x over a cube
    c2 = (lo * lo + lo * hi + hi * hi) / 3  # mean of x^2
    want = np.array([len(idx) * vol, *(c1.sum(0) * vol), (c2[:, 1] + c2[:, 2]).sum() * vol, (c2[:, 2] + c2[:, 0]).sum() * vol,
                     (c2[:, 0] + c2[:, 1]).sum() * vol, (c1[:, 0] * c1[:, 1]).sum() * vol, (c1[:, 1] * c1[:, 2]).sum() * vol,
                     (c1[:, 2] * c1[:, 0]).sum() * vol])
    np.testing.assert_allclose(got, want, rtol=2e-4)


def test_extraction_partitions_mass_momentum_and_angular_momentum():
    """determine_extracted_voxel_object_dynamics + apply_updated_inertial_properties_to_rigid_body (interaction.rs:405-585) on the
    two-sphere object (extraction.rs:2587-2624's scene): parent-after + fragment carry exactly the parent's mass, linear momentum
    and angular momentum about the old centre of mass; both keep the angular velocity; positions are the parts' centres of mass"""
    ext = 0.25
    o = ol.OracleObject.from_sdf(scenes.two_spheres_scene(25.0, 60.0), ext, 0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    m_all = o.inertia()[0]
    props = np.zeros(22, dtype=f32)
    ol.lib().orc_derive_inertial_properties(ol._p(m_all), ol._p(props))
    com0 = props[1:4].copy()
    I0 = props[4:13].reshape(3, 3).T.astype(np.float64)
    q = np.array([0.2, -0.1, 0.3, 0.0], dtype=np.float64)
    q[3] = np.sqrt(1 - q[:3] @ q[:3])
    body = ol.rigid_body_new(float(props[0]), I0, (1.0, 2.0, -0.5), velocity=(0.3, -0.2, 0.1), angular_velocity=(0.4, 1.1, -0.7), orientation=q.astype(f32))
    rc, child, origin = o.split_off_smallest_region()
    assert rc == 1
    m_child_own = child.inertia()[0]  # about the child's own grid origin
    m_child_in_parent = ol.offset_reference_point(m_child_own, -np.array(origin, dtype=f32) * f32(ext))
    m_parent_after = o.inertia()[0]
    np.testing.assert_allclose(m_parent_after + m_child_in_parent, m_all, rtol=3e-5, atol=1e-3)
    frag, m_back, com_frag = ol.extracted_object_dynamics(m_child_in_parent, origin, ext, com0, body)
    np.testing.assert_allclose(m_back, m_child_own, rtol=3e-4, atol=2e-2)  # there and back again
    parent, com_parent = ol.apply_updated_inertial_properties(body, m_parent_after, com0)
    M, X, V, Iw, W, L = state(body)
    parts = [state(parent), state(frag)]
    assert abs(sum(p[0] for p in parts) - M) <= 1e-5 * M
    np.testing.assert_allclose(sum(p[0] * p[2] for p in parts), M * V, rtol=1e-5, atol=1e-5 * M)
    np.testing.assert_allclose(sum(p[0] * p[1] for p in parts) / M, X, rtol=1e-5, atol=1e-5)  # the centre of mass stays where it was
    total_L = sum(p[5] + p[0] * np.cross(p[1] - X, p[2] - V) for p in parts)
    np.testing.assert_allclose(total_L, L, rtol=2e-5, atol=2e-5 * np.linalg.norm(L))
    for p in parts:
        np.testing.assert_allclose(p[4], W, rtol=2e-5, atol=2e-5)  # same angular velocity
        np.testing.assert_allclose(p[2], V + np.cross(W, p[1] - X), rtol=2e-5, atol=2e-5)  # rigid motion of the old body
    # the parts' local centres of mass are those of their own moments
    np.testing.assert_allclose(com_parent, m_parent_after[1:4] / m_parent_after[0], rtol=1e-6)
    np.testing.assert_allclose(com_frag, m_child_own[1:4] / m_child_own[0], rtol=1e-4)
    # momentum-preserving variant: only position and inertial properties change
    kept, _ = ol.apply_updated_inertial_properties(body, m_parent_after, com0, preserve_momentum=True)
    np.testing.assert_array_equal(kept["momentum"], body["momentum"])
    np.testing.assert_array_equal(kept["angular_momentum"], body["angular_momentum"])
    np.testing.assert_array_equal(kept["position"], parent["position"])
    assert kept["mass"] == parent["mass"]
