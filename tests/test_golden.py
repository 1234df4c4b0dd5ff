"""Committed golden vectors (tests/golden/*.json, made by tests/golden/make_golden.py): the oracle must still reproduce them
bit for bit (CPU), and the HIP path must match them (GPU) — voxel bytes, flags, labels, chunk records, mesh buffers as digests,
moments and rigid-body state within the tolerances of BASELINE.json's north_star (1e-5 relative)."""
import numpy as np
import pytest

import golden_util as gu


@pytest.fixture(scope="module")
def voxel_golden():
    return gu.load(gu.VOXEL_GOLDEN)["scenes"]


@pytest.fixture(scope="module")
def physics_golden():
    return gu.load(gu.PHYSICS_GOLDEN)["cases"]


@pytest.mark.parametrize("name", sorted(gu.scenes_small()))
def test_oracle_reproduces_voxel_golden(name, voxel_golden):
    gu.assert_digest_equal(gu.oracle_voxel_digest(gu.scenes_small()[name]), voxel_golden[name], moments_exact=True)


def test_oracle_reproduces_physics_golden(physics_golden):
    import oracle_lib as ol
    from impact_amd import scenes

    want = physics_golden["pile_4"]
    bodies, contacts = scenes.sphere_pile_scene(4)
    o = ol.OraclePhysics(bodies, None, tuple(want["config"]))
    for _ in range(want["steps"]):
        o.step(contacts, want["dt"])
    d, _ = o.bodies()
    for f in ("position", "orientation", "momentum", "angular_momentum"):
        got = [[float(x).hex() for x in row] for row in d[f]]
        assert got == want[f], f


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(gu.scenes_small()))
def test_hip_matches_voxel_golden(name, ctx, voxel_golden):
    gu.assert_digest_equal(gu.gpu_voxel_digest(ctx, gu.scenes_small()[name]), voxel_golden[name], moments_exact=False)


@pytest.mark.gpu
def test_hip_matches_physics_golden(ctx, physics_golden):
    from impact_amd import scenes
    from impact_amd.capi import CONTACT_DTYPE
    from impact_amd.physics import ConstraintSolverConfig, PhysicsWorld

    want = physics_golden["pile_4"]
    bodies, contacts = scenes.sphere_pile_scene(4)
    w = PhysicsWorld(ctx, ConstraintSolverConfig(*want["config"]))
    w.set_bodies(bodies, None)
    for _ in range(want["steps"]):
        w.perform_physics_step(np.ascontiguousarray(contacts, dtype=CONTACT_DTYPE), want["dt"])
    d = w.bodies()[0]
    for f in ("position", "orientation", "momentum", "angular_momentum"):
        ref = np.array([[float.fromhex(x) for x in row] for row in want[f]])
        got = d[f].astype(np.float64)
        scale = np.maximum(np.linalg.norm(ref, axis=1, keepdims=True), max(float(np.abs(ref).max()), 1e-30) * 1e-2)
        assert (np.abs(got - ref) / scale).max() <= 1e-5, f
    w.close()


def test_oracle_reproduces_next_rows_golden():
    want = gu.load(gu.NEXT_GOLDEN)
    assert want["script"] == gu.next_rows_script()
    gu.assert_next_rows_equal(gu.oracle_next_rows_digest(), want["digest"], exact=True)


@pytest.mark.gpu
def test_hip_matches_next_rows_golden(ctx):
    gu.assert_next_rows_equal(gu.gpu_next_rows_digest(ctx), gu.load(gu.NEXT_GOLDEN)["digest"], exact=False)
