"""Shared helpers for the rigid-body / contact-solver parity tests (GPU path vs oracle)."""
import numpy as np

import oracle_lib as ol
from impact_amd.capi import CONTACT_DTYPE, KINEMATIC_BODY_DTYPE
from impact_amd.physics import ConstraintSolverConfig, PhysicsWorld

STATE_FIELDS = ("position", "orientation", "momentum", "angular_momentum")
RTOL = 1e-5  # BASELINE.json north_star: rigid-body state within 1e-5 relative


def assert_bodies_close(gpu_dyn, orc_dyn, rtol=RTOL, what=""):
    """every state field within rtol of the oracle, relative to the field's magnitude (per-body values of a
    vector quantity are compared against that body's vector norm, floored by the field's typical size)"""
    for f in STATE_FIELDS:
        g = gpu_dyn[f].astype(np.float64)
        o = orc_dyn[f].astype(np.float64)
        norm = np.linalg.norm(o, axis=1, keepdims=True)
        floor = max(float(np.abs(o).max()), 1e-30) * 1e-2
        scale = np.maximum(norm, floor)
        err = np.abs(g - o) / scale
        assert err.max() <= rtol, f"{what}{f}: max rel err {err.max():.3e} at body {int(np.argmax(err.max(axis=1)))}"
    for f in ("mass", "inertia", "inv_inertia", "total_force", "total_torque"):
        np.testing.assert_array_equal(gpu_dyn[f], orc_dyn[f], err_msg=f)


def make_pair(ctx, dyn, kin=None, config=(8, 0.4, 3, 0.2)):
    w = PhysicsWorld(ctx, ConstraintSolverConfig(*config))
    w.set_bodies(dyn, kin)
    o = ol.OraclePhysics(dyn, kin, config)
    return w, o


def step_both(w, o, contacts, dt):
    c = np.ascontiguousarray(contacts, dtype=CONTACT_DTYPE)
    n_o = o.step(c, dt)
    r = w.perform_physics_step(c, dt)
    assert int(r["n_contacts"]) == n_o
    return r


def compare_contact_state(w, o, rtol=1e-4):
    ids, imp = w.contact_state()
    np.testing.assert_array_equal(ids, o.contact_order())
    oi = o.accumulated_impulses()
    scale = max(float(np.abs(oi).max()), 1e-30)
    assert np.abs(imp - oi).max() <= rtol * scale, (np.abs(imp - oi).max(), scale)


def static_plane():
    k = np.zeros(1, dtype=KINEMATIC_BODY_DTYPE)
    k["orientation"] = (0, 0, 0, 1)
    k["angular_axis"] = (0, 1, 0)
    return k


def smoke_check(ctx):
    """small pile step on the GPU vs the oracle (used by __graft_entry__.smoke)"""
    from impact_amd import scenes

    bodies, contacts = scenes.sphere_pile_scene(4)
    w, o = make_pair(ctx, bodies)
    for _ in range(2):
        step_both(w, o, contacts, 0.005)
    assert_bodies_close(w.bodies()[0], o.bodies()[0])
    w.close()
