"""Random mixtures of operations on one voxel body (fixed seeds): edits, contact queries against sphere / plane / capsule collidables with random
transforms, split-offs when the body falls apart, explicit range updates — the state one operation leaves (stale occupied ranges, converted
chunks, re-rooted regions) is the input of the next. After every operation the HIP path equals the oracle."""
import numpy as np
import pytest

import oracle_lib as ol
import parity_util as pu
import test_gpu_contacts as tcon
from impact_amd import scenes

pytestmark = pytest.mark.gpu
f32 = np.float32


def rand_q(rng):
    q = rng.normal(size=4)
    return (q / np.linalg.norm(q)).astype(f32)


@pytest.mark.parametrize("seed", pu.fuzz_seeds([11, 12, 13, 14, 15]))
def test_random_operation_mix(ctx, seed):
    rng = np.random.default_rng(seed)
    ext = [1.0, 0.5, 0.25][seed % 3]
    graph = [scenes.asteroid_scene(0.28), scenes.two_spheres_scene(14.0, 26.0), scenes.box_scene((36.0, 22.0, 28.0))][seed % 3]
    o = pu.oracle_from_graph(graph, ext)
    g = pu.gpu_from_graph(ctx, graph, ext)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    g.compute_all_derived_state()
    g.update_occupied_voxel_ranges()
    g.label_regions()
    ops_done = {k: 0 for k in ("sphere", "capsule", "csphere", "cplane", "ccapsule", "split", "ranges")}
    for step in range(16):
        occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float64)
        lo, hi = occ[:, 0], occ[:, 1]
        p_norm = (lo + rng.uniform(-0.05, 1.05, 3) * (hi - lo)).astype(f32)
        kind = rng.choice(["sphere", "capsule", "csphere", "cplane", "ccapsule", "split", "ranges"], p=[0.25, 0.2, 0.15, 0.1, 0.15, 0.1, 0.05])
        if kind in ("sphere", "capsule"):
            r = float(f32(rng.uniform(2.0, 8.0)))
            if kind == "sphere":
                ro, rg = o.absorb_sphere(p_norm, r + 2.0, r), g.absorb_sphere(p_norm, r + 2.0, r)
            else:
                v = (rng.normal(size=3) * rng.uniform(0.0, 20.0)).astype(f32)
                ro, rg = o.absorb_capsule(p_norm, v, r + 2.0, r), g.absorb_capsule(p_norm, v, r + 2.0, r)
            assert rg["touched_chunks"] == ro["touched_chunks"] and rg["removed_chunks"] == ro["removed_chunks"], (step, kind)
            np.testing.assert_array_equal(rg["emptied_by_type"], ro["emptied_by_type"])
            np.testing.assert_array_equal(rg["invalidated"], ro["invalidated"])
            pu.assert_edited_objects_equal(o, g, what=f"step {step} {kind}: ")
        elif kind in ("csphere", "cplane", "ccapsule"):
            q, t = rand_q(rng), rng.normal(size=3).astype(f32) * f32(3.0)
            qi = np.array([-q[0], -q[1], -q[2], q[3]], dtype=np.float64)
            x, y, z, w = qi
            b = np.array([x, y, z])
            v = p_norm.astype(np.float64) * ext - t.astype(np.float64)
            c_world = (v * (w * w - b @ b) + b * (2 * (v @ b)) + np.cross(b, v) * (2 * w)).astype(f32)  # a world point near the body
            resp = (0.3, 0.6, 0.5)
            if kind == "csphere":
                R = float(f32(rng.uniform(0.5, 6.0) * ext))
                want = tcon.oracle_contact_list(o, q, t, c_world, R, 5, 6, 0, 1, resp)
                got = g.sphere_contacts(q, t, c_world, R, 5, 6, 0, 1, resp)
            elif kind == "cplane":
                n = rng.normal(size=3)
                n = (n / np.linalg.norm(n)).astype(f32)
                disp = float(f32(np.dot(n.astype(np.float64), c_world.astype(np.float64))))
                want = tcon.oracle_plane_contact_list(o, q, t, n, disp, 5, 6, 0, 0x80000000, resp)
                got = g.plane_contacts(q, t, n, disp, 5, 6, 0, 0x80000000, resp)
            else:
                vec = (rng.normal(size=3) * rng.uniform(0.0, 10.0) * ext).astype(f32)
                R = float(f32(rng.uniform(0.5, 4.0) * ext))
                want = tcon.oracle_capsule_contact_list(o, q, t, c_world, vec, R, 5, 6, 0, 1, resp)
                got = g.capsule_contacts(q, t, c_world, vec, R, 5, 6, 0, 1, resp)
            tcon.assert_contacts_equal(got, want)
        elif kind == "split":
            if o.region_labels(False)[0] < 2:
                continue
            rc_o, child_o, origin_o = o.split_off_smallest_region()
            rc_g, child_g, origin_g, _ = g.extract_any_disconnected_region()
            assert rc_o == rc_g and (rc_o != 1 or tuple(int(x) for x in origin_g) == tuple(origin_o))  # the origin is the child's: only outcome 1 has one
            pu.assert_edited_objects_equal(o, g, what=f"step {step} split parent: ")
            if rc_o == 1:
                pu.assert_edited_objects_equal(child_o, child_g, what=f"step {step} split child: ")
                child_g.close()
        else:
            o.update_occupied_voxel_ranges()
            tight = g.update_occupied_voxel_ranges()
            assert [tuple(t) for t in tight] == [tuple(t) for t in o.info()["occupied_voxel_ranges"]]
        ops_done[kind] += 1
    assert sum(ops_done.values()) >= 10 or pu.fuzzing()
    g.close()
