"""Pins the oracle's rigid-body / sequential-impulses restatement (oracle/src/orc_physics.cpp) against the
reference's own constraint tests, engine/crates/impact_physics/tests/constraint.rs:246-576, and the
rigid-body unit tests rigid_body.rs:1096-1275, re-typed case by case with the same tolerances."""
import numpy as np
import pytest

import oracle_lib as ol
from impact_amd.capi import CONTACT_DTYPE, KINEMATIC_BIT, KINEMATIC_BODY_DTYPE


def contact(cid, a, b, geom, restitution, mu_s=0.0, mu_d=0.0, first=True):
    c = np.zeros((), dtype=CONTACT_DTYPE)
    c["id"], c["body_a"], c["body_b"] = cid, a, b
    c["position"], c["normal"], c["depth"] = geom
    c["restitution"], c["static_friction"], c["dynamic_friction"] = restitution, mu_s, mu_d
    c["flags"] = 1 if first else 0
    return c


def run_spheres(spheres, config, planes=()):
    """setup_bodies_and_run_constraints (tests/constraint.rs:209-244): spheres = (centre, radius, velocity,
    density, restitution); collidable sphere radius = the given radius, inertia = of_uniform_sphere(0.5, density)"""
    dyn = [ol.uniform_sphere_body(0.5, s[3], s[0], s[2]) for s in spheres]
    kin = np.zeros(len(planes), dtype=KINEMATIC_BODY_DTYPE)
    for k in kin:
        k["orientation"] = (0, 0, 0, 1)
        k["angular_axis"] = (0, 1, 0)
    contacts = []
    for i in range(len(spheres)):
        for j in range(i + 1, len(spheres)):
            g = ol.sphere_sphere_contact(spheres[i][0], spheres[i][1], spheres[j][0], spheres[j][1])
            if g is not None:
                # ContactResponseParameters::combined (material.rs:43-52): max restitution, sqrt(mu mu)
                contacts.append(contact(1000 * i + j, i, j, g, max(spheres[i][4], spheres[j][4])))
        for p, plane_restitution in enumerate(planes):
            g = ol.sphere_plane_contact(spheres[i][0], spheres[i][1])
            if g is not None:
                contacts.append(contact(5000 + 10 * i + p, i, KINEMATIC_BIT | p, g, max(spheres[i][4], plane_restitution)))
    w = ol.OraclePhysics(np.array(dyn), kin, config)
    cs = np.array(contacts, dtype=CONTACT_DTYPE) if contacts else np.zeros(0, dtype=CONTACT_DTYPE)
    n = w.prepare(cs)
    w.solve()
    return w, n


DEFAULT = (8, 0.4, 3, 0.2)
ONE_ITER = (1, 0.4, 0, 0.2)


def test_separated_bodies_unaffected_by_contact_constraints():
    """tests/constraint.rs:246-284"""
    spheres = [((x, 0.0, 0.0), 1.0, (0.0, 0.0, 0.0), 1.0, 1.0) for x in (0.0, 2.1)]
    w, n = run_spheres(spheres, DEFAULT)
    assert n == 0 and w.prepared_body_count() == 0
    dyn, _ = w.bodies()
    for b, s in zip(dyn, spheres):
        np.testing.assert_array_equal(b["position"], np.float32(s[0]))
        v, om = ol.body_motion(b)
        assert (v == 0).all() and (om == 0).all()


def binary_collision(sa, sb, va, vb):
    """test_binary_sphere_collision (tests/constraint.rs:286-337)"""
    w, n = run_spheres([sa, sb], ONE_ITER)
    assert n == 1 and w.prepared_body_count() == 2
    dyn, _ = w.bodies()
    for b, s, ve in zip(dyn, (sa, sb), (va, vb)):
        np.testing.assert_array_equal(b["position"], np.float32(s[0]))
        np.testing.assert_array_equal(b["orientation"], np.float32([0, 0, 0, 1]))
        v, om = ol.body_motion(b)
        np.testing.assert_allclose(v, ve, rtol=0, atol=1e-6)
        assert np.linalg.norm(om) <= 1e-6


def test_moving_sphere_colliding_head_on_with_same_mass_stationary_sphere():
    """tests/constraint.rs:339-367"""
    binary_collision(((0, 0, 0), 1.0, (0.5, 0, 0), 1.0, 1.0), ((2.0 - 1e-6, 0, 0), 1.0, (0, 0, 0), 1.0, 1.0), (0, 0, 0), (0.5, 0, 0))


def test_moving_sphere_colliding_head_on_with_very_massive_stationary_sphere():
    """tests/constraint.rs:369-396"""
    binary_collision(((0, 0, 0), 1.0, (0.5, 0, 0), 1.0, 1.0), ((2.0 - 1e-6, 0, 0), 1.0, (0, 0, 0), 1e9, 1.0), (-0.5, 0, 0), (0, 0, 0))


def test_moving_sphere_colliding_head_on_with_inelastic_same_mass_stationary_sphere():
    """tests/constraint.rs:398-426"""
    binary_collision(((0, 0, 0), 1.0, (0.5, 0, 0), 1.0, 0.0), ((2.0 - 1e-6, 0, 0), 1.0, (0, 0, 0), 1.0, 0.0), (0.25, 0, 0), (0.25, 0, 0))


def test_grazing_sphere_collision():
    """tests/constraint.rs:428-457: offset sqrt(2) r gives a 90 degree deflection"""
    off = float(np.float32(np.sqrt(np.float32(2.0))))
    binary_collision(((1e-6, 0, 0), 1.0, (0.5, 0, 0), 1.0, 1.0), ((off, off, 0), 1.0, (-0.5, 0, 0), 1.0, 1.0), (0, -0.5, 0), (0, 0.5, 0))


def test_sphere_colliding_with_static_plane():
    """tests/constraint.rs:459-514"""
    s = ((0.0, 1.0 - 1e-6, 0.0), 1.0, (0.5, -0.6, 0.0), 1.0, 1.0)
    w, n = run_spheres([s], ONE_ITER, planes=(1.0,))
    assert n == 1 and w.prepared_body_count() == 2
    dyn, _ = w.bodies()
    np.testing.assert_array_equal(dyn[0]["position"], np.float32(s[0]))
    v, om = ol.body_motion(dyn[0])
    np.testing.assert_allclose(v, (0.5, 0.6, 0.0), rtol=0, atol=1e-6)
    assert np.linalg.norm(om) <= 1e-6


def test_position_correction_of_interpenetrating_spheres():
    """tests/constraint.rs:516-576"""
    pen = 0.2
    spheres = [((0.5 * pen, 0, 0), 1.0, (0, 0, 0), 1.0, 1.0), ((2.0 - 0.5 * pen, 0, 0), 1.0, (0, 0, 0), 1.0, 1.0)]
    w, n = run_spheres(spheres, (0, 0.4, 1, 1.0))
    assert n == 1 and w.prepared_body_count() == 2
    dyn, _ = w.bodies()
    for idx, b in enumerate(dyn):
        np.testing.assert_allclose(b["position"], (2.0 * idx, 0, 0), rtol=0, atol=1e-6)
        np.testing.assert_array_equal(b["orientation"], np.float32([0, 0, 0, 1]))
        v, om = ol.body_motion(b)
        assert (v == 0).all() and np.linalg.norm(om) == 0


# ---- rigid_body.rs unit tests ---------------------------------------------------------------------
def dummy_body(velocity=(0, 0, 0), angular_velocity=(0, 0, 0)):
    return ol.rigid_body_new(1.0, np.eye(3), (0, 0, 0), velocity=velocity, angular_velocity=angular_velocity)


def test_should_retain_dynamic_body_velocities_when_advancing_for_zero_time_or_zero_force():
    """rigid_body.rs:1165-1214"""
    b = dummy_body((0, 0, 1), (1, 0, 0))
    b["total_force"] = (1, 0, 0)
    b["total_torque"] = np.cross((0, 1, 0), (1, 0, 0))
    w = ol.OraclePhysics(np.array([b]))
    w.advance_momenta(0.0)
    v, om = ol.body_motion(w.bodies()[0][0])
    np.testing.assert_allclose(v, (0, 0, 1), atol=1e-9)
    np.testing.assert_allclose(om, (1, 0, 0), atol=1e-6)
    w = ol.OraclePhysics(np.array([dummy_body()]))
    w.advance_momenta(1.0)
    v, om = ol.body_motion(w.bodies()[0][0])
    assert (v == 0).all() and (om == 0).all()


def test_should_change_dynamic_body_velocities_with_nonzero_force_and_torque():
    """rigid_body.rs:1216-1245"""
    b = dummy_body((0, 0, 1), (1, 0, 0))
    b["total_force"] = (1, 0, 0)
    b["total_torque"] = np.cross((0, 1, 0), (1, 0, 0))
    w = ol.OraclePhysics(np.array([b]))
    w.advance_momenta(1.0)
    v, om = ol.body_motion(w.bodies()[0][0])
    np.testing.assert_allclose(v, (1, 0, 1), atol=1e-6)
    np.testing.assert_allclose(om, (1, 0, -1), atol=1e-6)


def quat_axis_angle(q):
    n = np.linalg.norm(q[:3])
    return q[:3] / n, 2.0 * np.arctan2(n, q[3])


def test_advancing_orientation():
    """rigid_body.rs:1247-1275: zero speed / zero duration keep the orientation; rotation about the
    orientation's own axis adds angles (1e-8 in the reference; f32 here allows 1e-6)"""
    for om, dt in (((0, 0, 0), 1.2), ((1.2, 0, 0), 0.0)):
        w = ol.OraclePhysics(np.array([dummy_body(angular_velocity=om)]))
        w.advance_configurations(dt)
        np.testing.assert_allclose(w.bodies()[0][0]["orientation"], (0, 0, 0, 1), atol=1e-7)
    q0 = np.array([0, np.sin(0.05), 0, np.cos(0.05)], dtype=np.float32)
    b = ol.rigid_body_new(1.0, np.eye(3), (0, 0, 0), orientation=q0, angular_velocity=(0, 0.1, 0))
    w = ol.OraclePhysics(np.array([b]))
    w.advance_configurations(2.0)
    axis, ang = quat_axis_angle(w.bodies()[0][0]["orientation"].astype(np.float64))
    assert abs(ang - (0.1 + 0.1 * 2.0)) < 1e-6
    np.testing.assert_allclose(axis, (0, 1, 0), atol=1e-6)


# ---- solver bookkeeping -----------------------------------------------------------------------------
def test_constraint_cache_order_and_warm_start():
    """solver.rs:386-452: known ids keep their slot (and 0.4 x their impulses when the frame still
    matches), new ids are appended, ids not prepared again are swap-removed front to back."""
    dyn = np.array([ol.uniform_sphere_body(0.5, 1.0, (0.9 * i, 0, 0)) for i in range(5)])
    w = ol.OraclePhysics(dyn, config=DEFAULT)

    def pair(i, cid):
        g = ol.sphere_sphere_contact(dyn[i]["position"], 0.5, dyn[i + 1]["position"], 0.5)
        return contact(cid, i, i + 1, g, 0.0, 0.5, 0.5)

    w.prepare(np.array([pair(0, 10), pair(1, 11), pair(2, 12), pair(3, 13)]))
    np.testing.assert_array_equal(w.contact_order(), [10, 11, 12, 13])
    # give the spheres closing velocities so that impulses accumulate
    d2, _ = w.bodies()
    w.solve()
    w.prepare(np.array([pair(3, 13), pair(1, 11), pair(0, 99)]))
    # 10 and 12 vanish: idx0 (10) <- last (99); idx2 (12) <- last... order becomes [99, 11, 13]
    np.testing.assert_array_equal(w.contact_order(), [99, 11, 13])


def test_interlocked_manifold_is_replaced_by_one_separating_contact():
    """contact.rs:610-689: opposing penetration vectors -> one synthetic contact along the axis of least
    separation, infinite friction, zero restitution"""
    dyn = np.array([ol.uniform_sphere_body(0.5, 1.0, (0, 0.3, 0)), ol.uniform_sphere_body(0.5, 1.0, (0, 0, 0))])
    pts = [(-1.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.1, 1.0), (0.0, -0.1, -1.0)]
    nrm = [(1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, 0, -1)]
    cs = np.array([contact(20 + k, 0, 1, (pts[k], nrm[k], 0.1), 0.5, 0.7, 0.5, first=(k == 0)) for k in range(4)])
    w = ol.OraclePhysics(dyn, config=DEFAULT)
    assert w.prepare(cs) == 1
    # not interlocked when all normals agree
    cs2 = cs.copy()
    cs2["normal"] = (0, 1, 0)
    w2 = ol.OraclePhysics(dyn, config=DEFAULT)
    assert w2.prepare(cs2) == 4


def test_spherical_joint_is_the_placeholder_the_reference_has():
    """constraint/spherical_joint.rs:62-88: zero impulse, empty apply / positional correction. A joint therefore only makes its bodies
    constrained bodies (prepared-body count, velocity write-back = one f32 round trip of the momenta); positions, orientations and everything
    about bodies outside joints and contacts stay as they were"""
    dyn = np.array([ol.uniform_sphere_body(0.5, 1.7 + 0.3 * i, (3.0 * i, 0.0, 0.0), (0.123 + i, -0.456, 0.789)) for i in range(4)])
    w = ol.OraclePhysics(dyn, None, DEFAULT)
    w.set_spherical_joints([(0, 1)])
    n = w.prepare(np.zeros(0, dtype=CONTACT_DTYPE))
    assert n == 0 and w.prepared_body_count() == 2
    before = w.bodies()[0].copy()
    w.solve()
    after = w.bodies()[0]
    np.testing.assert_array_equal(after["position"], before["position"])
    np.testing.assert_array_equal(after["orientation"], before["orientation"])
    np.testing.assert_array_equal(after["momentum"][2:], before["momentum"][2:])  # not in a joint: untouched, to the bit
    m = before["mass"][:2, None]
    np.testing.assert_allclose(after["momentum"][:2], before["momentum"][:2], rtol=3e-7)  # mass * (momentum / mass)
    expect = (m * (before["momentum"][:2] * (np.float32(1.0) / m).astype(np.float32)).astype(np.float32)).astype(np.float32)
    assert np.all(np.abs(after["momentum"][:2] - expect) <= np.abs(expect) * 2e-7)
    # joints stay in force for the following steps; an empty list removes them
    w.set_spherical_joints([])
    w.prepare(np.zeros(0, dtype=CONTACT_DTYPE))
    assert w.prepared_body_count() == 0
