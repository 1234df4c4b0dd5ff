"""GPU parity of the rigid-body step + sequential-impulses contact solve (through the ivx_world_* C ABI)
against the oracle, plus the reference's own sphere-collision outcomes (impact_physics/tests/constraint.rs)
checked directly on the GPU results. Bar: body state within 1e-5 relative; ContactID order identical."""
import numpy as np
import pytest

import oracle_lib as ol
import physics_util as pu
from impact_amd import scenes
from impact_amd.capi import CONTACT_DTYPE, KINEMATIC_BIT
from impact_amd.physics import ConstraintSolverConfig, PhysicsWorld, uniform_sphere_body

pytestmark = pytest.mark.gpu


def contact(cid, a, b, geom, restitution, mu_s=0.0, mu_d=0.0, first=True):
    c = np.zeros((), dtype=CONTACT_DTYPE)
    c["id"], c["body_a"], c["body_b"] = cid, a, b
    c["position"], c["normal"], c["depth"] = geom
    c["restitution"], c["static_friction"], c["dynamic_friction"] = restitution, mu_s, mu_d
    c["flags"] = 1 if first else 0
    return c


def velocity(b):
    return b["momentum"] / b["mass"]


def binary(ctx, sa, sb, va, vb, config=(1, 0.4, 0, 0.2)):
    dyn = np.array([uniform_sphere_body(0.5, s[3], s[0], s[2]) for s in (sa, sb)])
    g = ol.sphere_sphere_contact(sa[0], sa[1], sb[0], sb[1])
    cs = np.array([contact(7, 0, 1, g, max(sa[4], sb[4]))])
    w, o = pu.make_pair(ctx, dyn, config=config)
    assert w.prepare_constraints(cs) == 1 and o.prepare(cs) == 1
    w.compute_and_apply_constrained_state()
    o.solve()
    gd, _ = w.bodies()
    for b, s, ve in zip(gd, (sa, sb), (va, vb)):
        np.testing.assert_array_equal(b["position"], np.float32(s[0]))
        np.testing.assert_array_equal(b["orientation"], np.float32([0, 0, 0, 1]))
        np.testing.assert_allclose(velocity(b), ve, rtol=0, atol=1e-6)
        assert np.abs(b["angular_momentum"]).max() <= 1e-6
    pu.assert_bodies_close(gd, o.bodies()[0])
    w.close()


def test_reference_sphere_collisions(ctx):
    """tests/constraint.rs:339-457: head-on equal mass, very massive, inelastic, grazing"""
    binary(ctx, ((0, 0, 0), 1.0, (0.5, 0, 0), 1.0, 1.0), ((2.0 - 1e-6, 0, 0), 1.0, (0, 0, 0), 1.0, 1.0), (0, 0, 0), (0.5, 0, 0))
    binary(ctx, ((0, 0, 0), 1.0, (0.5, 0, 0), 1.0, 1.0), ((2.0 - 1e-6, 0, 0), 1.0, (0, 0, 0), 1e9, 1.0), (-0.5, 0, 0), (0, 0, 0))
    binary(ctx, ((0, 0, 0), 1.0, (0.5, 0, 0), 1.0, 0.0), ((2.0 - 1e-6, 0, 0), 1.0, (0, 0, 0), 1.0, 0.0), (0.25, 0, 0), (0.25, 0, 0))
    off = float(np.float32(np.sqrt(np.float32(2.0))))
    binary(ctx, ((1e-6, 0, 0), 1.0, (0.5, 0, 0), 1.0, 1.0), ((off, off, 0), 1.0, (-0.5, 0, 0), 1.0, 1.0), (0, -0.5, 0), (0, 0.5, 0))


def test_reference_sphere_on_static_plane(ctx):
    """tests/constraint.rs:459-514 (kinematic body = static plane)"""
    s = ((0.0, 1.0 - 1e-6, 0.0), 1.0, (0.5, -0.6, 0.0), 1.0, 1.0)
    dyn = np.array([uniform_sphere_body(0.5, s[3], s[0], s[2])])
    g = ol.sphere_plane_contact(s[0], s[1])
    cs = np.array([contact(9, 0, KINEMATIC_BIT | 0, g, 1.0)])
    w, o = pu.make_pair(ctx, dyn, pu.static_plane(), config=(1, 0.4, 0, 0.2))
    assert w.prepare_constraints(cs) == 1 and o.prepare(cs) == 1
    w.compute_and_apply_constrained_state()
    o.solve()
    gd, gk = w.bodies()
    np.testing.assert_allclose(velocity(gd[0]), (0.5, 0.6, 0.0), rtol=0, atol=1e-6)
    pu.assert_bodies_close(gd, o.bodies()[0])
    np.testing.assert_array_equal(gk["position"], 0)
    w.close()


def test_reference_position_correction(ctx):
    """tests/constraint.rs:516-576"""
    pen = 0.2
    spheres = [((0.5 * pen, 0, 0), 1.0), ((2.0 - 0.5 * pen, 0, 0), 1.0)]
    dyn = np.array([uniform_sphere_body(0.5, 1.0, s[0]) for s in spheres])
    g = ol.sphere_sphere_contact(spheres[0][0], 1.0, spheres[1][0], 1.0)
    cs = np.array([contact(3, 0, 1, g, 1.0)])
    w, o = pu.make_pair(ctx, dyn, config=(0, 0.4, 1, 1.0))
    w.prepare_constraints(cs)
    o.prepare(cs)
    w.compute_and_apply_constrained_state()
    o.solve()
    gd, _ = w.bodies()
    for idx, b in enumerate(gd):
        np.testing.assert_allclose(b["position"], (2.0 * idx, 0, 0), rtol=0, atol=1e-6)
        np.testing.assert_array_equal(b["orientation"], np.float32([0, 0, 0, 1]))
        assert (b["momentum"] == 0).all()
    pu.assert_bodies_close(gd, o.bodies()[0])
    w.close()


@pytest.mark.parametrize("n,steps", [(3, 4), (6, 3)])
def test_small_piles_step_by_step(ctx, n, steps):
    """lattice piles under gravity: several steps with the contact list handed over again every step
    (cache hits, warm starting with weight 0.4), state + contact order + impulses vs the oracle"""
    bodies, contacts = scenes.sphere_pile_scene(n)
    w, o = pu.make_pair(ctx, bodies)
    for s in range(steps):
        r = pu.step_both(w, o, contacts, 0.005)
        pu.assert_bodies_close(w.bodies()[0], o.bodies()[0], what=f"step {s}: ")
        pu.compare_contact_state(w, o)
    assert int(r["n_bodies"]) == n ** 3
    w.close()


def test_config4_pile_4096_bodies(ctx):
    """BASELINE config 4: 16^3 = 4096 spheres, 46 080 contacts, 8 + 3 iterations, dt 0.005"""
    bodies, contacts = scenes.sphere_pile_scene(16)
    assert len(bodies) == 4096 and len(contacts) == 46080
    w, o = pu.make_pair(ctx, bodies)
    for s in range(2):
        r = pu.step_both(w, o, contacts, 0.005)
        pu.assert_bodies_close(w.bodies()[0], o.bodies()[0], what=f"step {s}: ")
    pu.compare_contact_state(w, o)
    # a further step over the RESIDENT contact set (what bench.py times) = prepare_constraints again with the same list
    o.step(contacts, 0.005)
    w.step(0.005)
    pu.assert_bodies_close(w.bodies()[0], o.bodies()[0], what="resident step: ")
    assert int(r["n_levels"][0]) > 0 and int(r["n_levels"][1]) > 0
    w.close()


def test_config4_pile_60_steps(ctx):
    """BASELINE config 4 over 60 frames against the oracle (26 ms per oracle step): does 1e-5 survive the warm-start history?
    The scale a body's error is measured against is its own vector norm, floored at 1e-4 of the field's largest entry (the
    short test above floors at 1e-2)."""
    bodies, contacts = scenes.sphere_pile_scene(16)
    w, o = pu.make_pair(ctx, bodies)
    worst = 0.0
    for s in range(60):
        if s % 7 == 3:  # resident set: the device keeps last frame's contacts
            o.step(contacts, 0.005)
            w.step(0.005)
        else:
            pu.step_both(w, o, contacts, 0.005)
        if s in (0, 1, 9, 29, 59):
            gd, od = w.bodies()[0], o.bodies()[0]
            for f in pu.STATE_FIELDS:
                g64, o64 = gd[f].astype(np.float64), od[f].astype(np.float64)
                scale = np.maximum(np.linalg.norm(o64, axis=1, keepdims=True), max(float(np.abs(o64).max()), 1e-30) * 1e-4)
                err = float((np.abs(g64 - o64) / scale).max())
                worst = max(worst, err)
                assert err <= pu.RTOL, f"step {s} {f}: {err:.3e}"
            pu.compare_contact_state(w, o)
    print(f"config 4, 60 steps: worst relative error {worst:.3e}")
    w.close()


@pytest.mark.parametrize("groups", [2, 5, 16, 255])
def test_solve_on_several_workgroups_is_the_same_solve(ctx, groups):
    """the level schedule walked by `groups` workgroups (k_solve_mg; 255: the chain-stationary solve, k_solve_cs) against the single-workgroup
    kernel (bodies in LDS): bit-identical bodies and impulses, frame after frame, with contacts that come and go; and against the oracle"""
    rng = np.random.default_rng(11)
    bodies, contacts = scenes.sphere_pile_scene(6)
    bodies["momentum"] += rng.normal(0, 0.05, bodies["momentum"].shape).astype(np.float32)
    bodies["angular_momentum"] += rng.normal(0, 0.01, bodies["angular_momentum"].shape).astype(np.float32)
    w1, o = pu.make_pair(ctx, bodies)
    wg, _ = pu.make_pair(ctx, bodies)
    w1.set_solver_groups(1)
    wg.set_solver_groups(groups)
    manifolds = contacts.reshape(-1, 4)
    full, half = np.arange(len(manifolds)), np.arange(0, len(manifolds), 2)
    for s, keep in enumerate([full, full, half, full[::-1], full, half, full, full]):
        cs = manifolds[keep].reshape(-1).copy()
        pu.step_both(w1, o, cs, 0.004)
        wg.perform_physics_step(cs, 0.004)
        if groups == 255:
            assert wg.solver_info()["kernel"] == "chain_stationary" and w1.solver_info()["kernel"] == "one_workgroup"
        else:
            assert wg.solver_info()["workgroups"] == groups and w1.solver_info()["workgroups"] == 1
        d1, dg = w1.bodies()[0], wg.bodies()[0]
        for f in pu.STATE_FIELDS:
            np.testing.assert_array_equal(dg[f].view(np.uint32), d1[f].view(np.uint32), err_msg=f"frame {s} {f}")
        np.testing.assert_array_equal(wg.contact_state()[1].view(np.uint32), w1.contact_state()[1].view(np.uint32))
        pu.assert_bodies_close(dg, o.bodies()[0], what=f"frame {s}: ")
    w1.close()
    wg.close()


def test_contacts_come_and_go(ctx):
    """ConstraintCache order after removals (swap_remove) and additions, with spinning / moving bodies and
    friction; ragged inputs: empty contact list, bodies without contacts"""
    rng = np.random.default_rng(5)
    bodies, contacts = scenes.sphere_pile_scene(4)
    bodies["momentum"] += rng.normal(0, 0.05, bodies["momentum"].shape).astype(np.float32)
    bodies["angular_momentum"] += rng.normal(0, 0.01, bodies["angular_momentum"].shape).astype(np.float32)
    w, o = pu.make_pair(ctx, bodies)
    manifolds = contacts.reshape(-1, 4)
    keep_sets = [np.arange(len(manifolds)), np.arange(0, len(manifolds), 2), np.arange(len(manifolds))[::-1][:40], np.array([], dtype=int),
                 np.arange(5, len(manifolds))]
    for s, keep in enumerate(keep_sets):
        cs = manifolds[keep].reshape(-1) if len(keep) else np.zeros(0, dtype=CONTACT_DTYPE)
        pu.step_both(w, o, cs, 0.004)
        pu.assert_bodies_close(w.bodies()[0], o.bodies()[0], what=f"set {s}: ")
        if len(cs):
            pu.compare_contact_state(w, o)
    w.close()


def test_repeated_and_changing_contact_sets_alternate(ctx):
    """the steady-state shortcuts of ivx_world_set_contacts (same ids in the same order keep their slots without lookups; an unchanged chain
    structure keeps its schedule) between frames that change the set, reorder it, or only move the contact geometry: every frame against
    the oracle, contact order and warm-started impulses included"""
    rng = np.random.default_rng(11)
    bodies, contacts = scenes.sphere_pile_scene(4)
    bodies["momentum"] += rng.normal(0, 0.05, bodies["momentum"].shape).astype(np.float32)
    w, o = pu.make_pair(ctx, bodies)
    manifolds = contacts.reshape(-1, 4)
    full, half = np.arange(len(manifolds)), np.arange(0, len(manifolds), 2)
    frames = [full, full, full, half, half, full, full, full[::-1], full[::-1], half, half, half]
    for s, keep in enumerate(frames):
        cs = manifolds[keep].reshape(-1).copy()
        if s % 3 == 1:  # same ids and structure, other geometry: the schedule is reused, the contacts themselves are not
            cs["depth"] = (cs["depth"] * np.float32(0.5)).astype(np.float32)
        pu.step_both(w, o, cs, 0.004)
        pu.assert_bodies_close(w.bodies()[0], o.bodies()[0], what=f"frame {s}: ")
        pu.compare_contact_state(w, o)
    w.close()


def test_interlocked_manifold(ctx):
    """contact.rs:610-780 through the GPU path: opposing penetration vectors are replaced by one
    separating contact (needs the bodies' current positions from the device)"""
    dyn = np.array([uniform_sphere_body(0.5, 1.0, (0, 0.3, 0)), uniform_sphere_body(0.5, 1.0, (0, 0, 0))])
    pts = [(-1.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.1, 1.0), (0.0, -0.1, -1.0)]
    nrm = [(1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, 0, -1)]
    cs = np.array([contact(20 + k, 0, 1, (pts[k], nrm[k], 0.1), 0.5, 0.7, 0.5, first=(k == 0)) for k in range(4)])
    w, o = pu.make_pair(ctx, dyn)
    for _ in range(2):
        r = pu.step_both(w, o, cs, 0.01)
        assert int(r["n_contacts"]) == 1
        pu.assert_bodies_close(w.bodies()[0], o.bodies()[0])
        pu.compare_contact_state(w, o)
    w.close()


def test_free_bodies_integrate_exactly_like_the_oracle(ctx):
    """no contacts: momenta advance by force and torque, positions by velocity, orientations by the exact
    axis-angle rotation (rigid_body.rs:708-742, 1013-1034); anisotropic inertia, 20 steps"""
    rng = np.random.default_rng(1)
    n = 64
    dyn = []
    for i in range(n):
        I = np.diag(rng.uniform(0.5, 2.0, 3))
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        dyn.append(ol.rigid_body_new(rng.uniform(0.5, 3.0), I, rng.normal(size=3), q, rng.normal(size=3), rng.normal(size=3)))
    dyn = np.array(dyn)
    dyn["total_force"] = rng.normal(size=(n, 3)).astype(np.float32)
    dyn["total_torque"] = rng.normal(size=(n, 3)).astype(np.float32)
    w, o = pu.make_pair(ctx, dyn)
    none = np.zeros(0, dtype=CONTACT_DTYPE)
    for _ in range(20):
        pu.step_both(w, o, none, 0.01)
    pu.assert_bodies_close(w.bodies()[0], o.bodies()[0])
    # the contact list stays empty: `step` / `step_enqueue` without a new `prepare_constraints` take the single-launch path
    for k in range(10):
        o.step(none, 0.01)
        if k % 2:
            w.step(0.01)
        else:
            w.step_enqueue(0.01)
    pu.assert_bodies_close(w.bodies()[0], o.bodies()[0])
    w.close()


def test_spherical_joints_only_register_their_bodies(ctx):
    """The reference's SphericalJoint is a placeholder: no impulse, no positional correction (constraint/spherical_joint.rs:62-88). Preparing
    it makes its two bodies constrained bodies of the step — velocities synchronised before the solve and written back after it (momentum =
    mass x (momentum / mass), an f32 round trip) — and that is all. Jointed bodies without contacts, jointed bodies with contacts, a joint to a
    kinematic body, free bodies beside them: bit-equal to the oracle; the free bodies keep their momenta to the bit."""
    rng = np.random.default_rng(5)
    n = 24
    dyn = []
    for i in range(n):
        I = np.diag(rng.uniform(0.5, 2.0, 3))
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        dyn.append(ol.rigid_body_new(rng.uniform(0.37, 3.1), I, rng.normal(size=3) * 5, q, rng.normal(size=3), rng.normal(size=3)))
    dyn = np.array(dyn)
    from impact_amd.capi import KINEMATIC_BODY_DTYPE

    kin = np.zeros(1, dtype=KINEMATIC_BODY_DTYPE)
    kin["velocity"] = (0.5, 0.0, -0.25)
    kin["angular_axis"] = (0.0, 0.0, 1.0)
    w, o = pu.make_pair(ctx, dyn, kin)
    joints = [(0, 1), (1, 2), (5, 9), (11, KINEMATIC_BIT | 0)]
    w.set_spherical_joints(joints)
    o.set_spherical_joints(joints)
    none = np.zeros(0, dtype=CONTACT_DTYPE)
    before = w.bodies()[0].copy()
    for _ in range(5):
        pu.step_both(w, o, none, 0.01)
    got, want = w.bodies()[0], o.bodies()[0]
    for f in ("momentum", "angular_momentum", "position", "orientation"):
        np.testing.assert_array_equal(got[f].view(np.uint32), want[f].view(np.uint32), err_msg=f)
    free = [i for i in range(n) if i not in (0, 1, 2, 5, 9, 11)]
    np.testing.assert_array_equal(got["momentum"][free].view(np.uint32), before["momentum"][free].view(np.uint32))
    # with a contact in the same step
    c = np.array([contact(77, 2, 3, ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.01), 0.5)], dtype=CONTACT_DTYPE)
    for _ in range(3):
        pu.step_both(w, o, c, 0.01)
    pu.assert_bodies_close(w.bodies()[0], o.bodies()[0])
    w.close()


def test_world_and_voxel_step_on_two_contexts_overlap(ctx):
    """the rigid-body world on a context (HIP stream) of its own while a voxel object steps on another (bench.py's two-stream frame):
    the multi-workgroup solve shares the chip with the voxel kernels and both give what they give alone, bit for bit"""
    from impact_amd import capi
    from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject

    rng = np.random.default_rng(3)
    bodies, contacts = scenes.sphere_pile_scene(8)
    bodies["momentum"] += rng.normal(0, 0.05, bodies["momentum"].shape).astype(np.float32)
    ctx2 = Context(0)
    alone, beside = PhysicsWorld(ctx), PhysicsWorld(ctx2)
    for w in (alone, beside):
        w.set_bodies(bodies)
        w.prepare_constraints(contacts)
        w.set_solver_groups(4)
    gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(1.0), 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    want = obj.step(capi.STAGE_ALL).copy()
    for frame in range(4):
        alone.step(0.004)
        beside.step_enqueue(0.004)
        obj.step_enqueue(capi.STAGE_ALL)
        got = obj.step_collect()
        ctx2.synchronize()
        for f in ("region_count", "mesh", "moments", "occupied"):
            assert np.atleast_1d(got[f]).tobytes() == np.atleast_1d(want[f]).tobytes(), f"frame {frame} {f}"
        da, db = alone.bodies()[0], beside.bodies()[0]
        for f in pu.STATE_FIELDS:
            np.testing.assert_array_equal(db[f].view(np.uint32), da[f].view(np.uint32), err_msg=f"frame {frame} {f}")
    assert beside.solver_info()["workgroups"] == 4
    for w in (alone, beside):
        w.close()
    obj.close()
    ctx2.close()
