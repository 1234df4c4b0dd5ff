"""Random relative poses of two voxel bodies (fixed seeds): the overlap bounds of the two occupied boxes (compute_box_intersection_bounds has a
branch per edge and axis), the probe filter and the SDF probe are exercised with axis-aligned, slightly tilted and arbitrary rotations, touching,
overlapping and separated placements; the mutual contacts equal the oracle's bit for bit at every pose, and mutual absorption at some of them
leaves both objects equal to the oracle's."""
import numpy as np
import pytest

import oracle_lib as ol
import parity_util as pu
import test_gpu_collide as tc
from impact_amd import scenes
from impact_amd.voxel import VoxelObjectMesh

pytestmark = pytest.mark.gpu
f32 = np.float32


def random_quaternion(rng, kind):
    if kind == 0:
        return np.array([0, 0, 0, 1], dtype=f32)
    if kind == 1:  # a quarter turn about an axis: edges of one box parallel to faces of the other
        a = np.zeros(3)
        a[rng.integers(3)] = 1.0
        return np.array([*(a * np.sin(np.pi / 4)), np.cos(np.pi / 4)], dtype=f32)
    if kind == 2:  # nearly aligned
        a = rng.normal(size=3)
        a /= np.linalg.norm(a)
        return np.array([*(a * np.sin(5e-4)), np.cos(5e-4)], dtype=f32)
    q = rng.normal(size=4)
    return (q / np.linalg.norm(q)).astype(f32)


@pytest.mark.parametrize("seed", pu.fuzz_seeds([1, 2, 3, 4]))
def test_random_poses(ctx, seed):
    rng = np.random.default_rng(100 + seed)
    ga = scenes.asteroid_scene(0.3) if seed % 2 else scenes.box_scene((44.0, 20.0, 30.0))
    gb = scenes.sphere_scene(16.0) if seed < 3 else scenes.box_scene((18.0, 26.0, 12.0))
    ea, eb = (1.0, 1.0) if seed != 2 else (0.5, 1.0)
    A, GA = tc.both(ctx, ga, ea)
    B, GB = tc.both(ctx, gb, eb)
    pa, pb = tc.probes_both(A, GA), tc.probes_both(B, GB)
    ca = (np.array([0.5 * (a + b) for a, b in A.info()["occupied_voxel_ranges"]]) * ea).astype(f32)
    cb = (np.array([0.5 * (a + b) for a, b in B.info()["occupied_voxel_ranges"]]) * eb).astype(f32)
    ra = 0.5 * min(b - a for a, b in A.info()["occupied_voxel_ranges"]) * ea
    rb = 0.5 * min(b - a for a, b in B.info()["occupied_voxel_ranges"]) * eb
    n_with_contacts = 0
    for pose in range(10):
        qa, qb = random_quaternion(rng, pose % 4), random_quaternion(rng, (pose // 2) % 4)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        dist = (ra + rb) * rng.uniform(0.55, 1.25)
        ta = tc.placed(ca, qa, rng.normal(size=3))
        centre_a_world = tc.rot64(np.array([-qa[0], -qa[1], -qa[2], qa[3]], dtype=np.float64), ca.astype(np.float64) - ta.astype(np.float64))
        tb = tc.placed(cb, qb, centre_a_world + d * dist)
        want, wi = tc.oracle_contact_list(A, pa, ca, qa, ta, B, pb, cb, qb, tb, 7, 9, 0, 1, (0.1, 0.2, 0.3))
        got = GA.mutual_contacts(qa, ta, ca, GB, qb, tb, cb, 7, 9, 0, 1, (0.1, 0.2, 0.3))
        tc.assert_contacts_equal(got, want)
        n_with_contacts += len(want) > 0
    assert n_with_contacts >= 3 or pu.fuzzing()
    # and two rounds of mutual absorption at overlapping poses (the objects change in between)
    for rnd in range(2):
        qa, qb = random_quaternion(rng, 3), random_quaternion(rng, 2 + rnd)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        ta = tc.placed(ca, qa, [0.0, 0.0, 0.0])
        tb = tc.placed(cb, qb, d * (ra + rb) * 0.7)
        roa, rob = A.absorb_mutual(qa, ta, B, qb, tb, float(rnd))
        rga, rgb = GA.absorb_mutual(qa, ta, GB, qb, tb, float(rnd))
        for ro, rg, o, g in ((roa, rga, A, GA), (rob, rgb, B, GB)):
            assert rg["emptied_voxels"] == ro["emptied_voxels"] and rg["touched_chunks"] == ro["touched_chunks"]
            np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
            pu.assert_edited_objects_equal(o, g)
        assert roa["emptied_voxels"] + rob["emptied_voxels"] > 100 or pu.fuzzing()
    GA.close()
    GB.close()
