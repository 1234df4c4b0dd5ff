"""GPU parity of the contact generation between two voxel objects (second part): collision probes picked from
the mesh and the mutual contacts, HIP path through the C ABI against the oracle (pinned in tests/test_oracle_collide.py): probe
points and chunk ranges bit-exact, the same contacts in the same order with the same ids, geometry bit-exact; then two voxel
bodies dropped onto each other, stepped through contact generation -> solver -> integration against the oracle."""
import numpy as np
import pytest

import oracle_lib as ol
import parity_util as pu
from impact_amd import scenes
from impact_amd.capi import CONTACT_DTYPE
from impact_amd.voxel import VoxelObjectMesh

pytestmark = pytest.mark.gpu
f32 = np.float32


def both(ctx, graph, extent=1.0):
    o = pu.oracle_from_graph(graph, extent)
    g = pu.gpu_from_graph(ctx, graph, extent)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    g.compute_all_derived_state()
    g.update_occupied_voxel_ranges()
    g.label_regions()
    g.mesh = VoxelObjectMesh.create(g)
    return o, g


def probes_both(o, g):
    want_pts, want_ents = o.collision_probes(o.mesh())
    n = g.collision_probes_recompute()
    got_pts, got_ents = g.collision_probes()
    assert n == len(want_pts)
    np.testing.assert_array_equal(got_ents, want_ents)
    np.testing.assert_array_equal(got_pts.view(np.uint32), want_pts.view(np.uint32))
    return want_pts, want_ents


@pytest.mark.parametrize("case", ["sphere_block8", "small_block4", "thin_block2", "sliver_block1", "half_extent", "asteroid"])
def test_collision_probes(ctx, case):
    graph, extent = {
        "sphere_block8": (scenes.sphere_scene(30.0), 1.0),
        "small_block4": (scenes.box_scene((9.0, 12.0, 20.0)), 1.0),
        "thin_block2": (scenes.box_scene((5.0, 40.0, 18.0)), 1.0),
        "sliver_block1": (scenes.box_scene((3.0, 20.0, 18.0)), 1.0),
        "half_extent": (scenes.sphere_scene(11.0), 0.5),
        "asteroid": (scenes.asteroid_scene(0.4), 1.0),
    }[case]
    o, g = both(ctx, graph, extent)
    pts, ents = probes_both(o, g)
    assert len(pts) > 8
    g.close()


def oracle_contact_list(A, pa, ca, qa, ta, B, pb, cb, qb, tb, id_a, id_b, body_a, body_b, response):
    wi, pos, nrm, dep = A.mutual_contacts(pa, ca, qa, ta, B, pb, cb, qb, tb)
    out = np.zeros(len(wi), dtype=CONTACT_DTYPE)
    for n, (w, p, nn, d) in enumerate(zip(wi, pos, nrm, dep)):
        out[n]["id"] = scenes.contact_id(id_a, id_b, 0, int(w[1]), int(w[2]), int(w[3]))
        out[n]["body_a"], out[n]["body_b"] = body_a, body_b
        out[n]["position"], out[n]["normal"], out[n]["depth"] = p, nn, d
        out[n]["restitution"], out[n]["static_friction"], out[n]["dynamic_friction"] = response
        out[n]["flags"] = 1 if n == 0 else 0
    return out, wi


def assert_contacts_equal(got, want):
    assert len(got) == len(want)
    for f in ("id", "body_a", "body_b", "flags"):
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    for f in ("position", "normal", "depth", "restitution", "static_friction", "dynamic_friction"):
        np.testing.assert_array_equal(got[f].view(np.uint32), want[f].view(np.uint32), err_msg=f)


def rot64(q, v):
    x, y, z, w = [float(a) for a in q]
    b = np.array([x, y, z])
    return v * (w * w - b @ b) + b * (2 * (v @ b)) + np.cross(b, v) * (2 * w)


def placed(com, q, world_pos):
    """world -> object translation that puts the object's centre of mass at `world_pos` with orientation q^-1"""
    return (com.astype(np.float64) - rot64(q, np.asarray(world_pos, dtype=np.float64))).astype(f32)


@pytest.mark.parametrize("case", ["two_spheres", "sphere_into_box_rotated", "mixed_extents", "deep", "apart", "both_rotated"])
def test_mutual_contacts(ctx, case):
    spec = {
        "two_spheres": (scenes.sphere_scene(30.0), 1.0, scenes.sphere_scene(22.0), 1.0, 48.0, 0.0, 0.0),
        "sphere_into_box_rotated": (scenes.box_scene((80.0, 24.0, 80.0)), 1.0, scenes.sphere_scene(24.0), 1.0, 33.0, 0.0, 0.6),
        "mixed_extents": (scenes.sphere_scene(28.0), 0.5, scenes.sphere_scene(18.0), 1.0, 29.0, 0.0, -0.4),
        "deep": (scenes.sphere_scene(40.0), 1.0, scenes.sphere_scene(20.0), 1.0, 30.0, 0.0, 0.9),  # B reaches A's uniform core
        "apart": (scenes.sphere_scene(20.0), 1.0, scenes.sphere_scene(20.0), 1.0, 90.0, 0.2, 0.3),
        "both_rotated": (scenes.asteroid_scene(0.3), 1.0, scenes.box_scene((30.0, 30.0, 30.0)), 1.0, 38.0, 0.7, -1.1),
    }[case]
    ga, ea, gb, eb, sep, ang_a, ang_b = spec
    A, GA = both(ctx, ga, ea)
    B, GB = both(ctx, gb, eb)
    pa, pb = probes_both(A, GA), probes_both(B, GB)
    ca, cb = A.center_of_mass(), B.center_of_mass()
    ax_a = np.array([1.0, -0.2, 0.4]) / np.linalg.norm([1.0, -0.2, 0.4])
    ax_b = np.array([0.3, 0.1, 1.0]) / np.linalg.norm([0.3, 0.1, 1.0])
    qa = np.array([*(ax_a * np.sin(ang_a / 2)), np.cos(ang_a / 2)], dtype=f32)
    qb = np.array([*(ax_b * np.sin(ang_b / 2)), np.cos(ang_b / 2)], dtype=f32)
    ta = placed(ca, qa, [0.5, -0.25, 0.125])
    tb = placed(cb, qb, [0.5, -0.25 + sep, 0.125])
    resp = (0.3, 0.6, 0.45)
    want, wi = oracle_contact_list(A, pa, ca, qa, ta, B, pb, cb, qb, tb, 11, 22, 0, 1, resp)
    got = GA.mutual_contacts(qa, ta, ca, GB, qb, tb, cb, 11, 22, 0, 1, resp)
    if case == "apart":
        assert len(want) == 0
    else:
        assert len(want) > 10 and set(wi[:, 0].tolist()) == {0, 1}
    if case == "deep":
        assert np.isclose(want["depth"], 2.56).sum() > 3  # the capped distance with the centre-of-mass direction
    assert_contacts_equal(got, want)
    GA.close()
    GB.close()


def test_mutual_contacts_of_many_pairs(ctx):
    """the pairs list (ivx_mutual_voxel_object_contacts_many) against the single-pair call: a row of five bodies of different shapes and extents,
    neighbours touching, one pair far apart, one body in three pairs; the manifolds must be the single-pair lists, byte for byte"""
    from impact_amd import many
    from impact_amd.capi import IvxError

    shapes = [(scenes.sphere_scene(24.0), 1.0), (scenes.box_scene((30.0, 30.0, 30.0)), 1.0), (scenes.sphere_scene(28.0), 0.5), (scenes.asteroid_scene(0.3), 1.0),
              (scenes.sphere_scene(16.0), 1.0)]
    objs = [both(ctx, g, e) for g, e in shapes]
    probes = [probes_both(o, g) for o, g in objs]
    rng = np.random.default_rng(5)
    pose = []
    x = 0.0
    for i, (o, g) in enumerate(objs):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = rng.uniform(-1.0, 1.0)
        q = np.array([*(ax * np.sin(ang / 2)), np.cos(ang / 2)], dtype=f32)
        com = o.center_of_mass()
        pose.append((q, placed(com, q, [x, 0.25, -0.5]), com))
        x += 24.0
    pair_ids = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 4), (1, 3), (2, 1)]  # (0, 4): apart; body 1 in four pairs, once as B
    resp = (0.2, 0.5, 0.4)
    pairs, want = [], []
    for n, (i, j) in enumerate(pair_ids):
        (qa, ta, ca), (qb, tb, cb) = pose[i], pose[j]
        pairs.append(dict(a=objs[i][1], b=objs[j][1], rotation_a=qa, translation_a=ta, center_of_mass_a=ca, rotation_b=qb, translation_b=tb, center_of_mass_b=cb,
                          collidable_id_a=100 + i, collidable_id_b=100 + j, body_a=i, body_b=j, response=resp))
        want.append(objs[i][1].mutual_contacts(qa, ta, ca, objs[j][1], qb, tb, cb, 100 + i, 100 + j, i, j, resp))
    assert sum(len(w) > 10 for w in want) >= 4 and len(want[4]) == 0
    got, off = many.mutual_voxel_object_contacts_many(many.mutual_queries(pairs))
    assert off[0] == 0 and off[-1] == len(got) == sum(len(w) for w in want)
    for n, w in enumerate(want):
        assert_contacts_equal(got[off[n] : off[n + 1]], w)
    # ... and against the ORACLE's lists (the batched form's scan / count / emit code is the single-pair call's: a defect the two share would
    # pass the comparison above)
    for n, (i, j) in enumerate(pair_ids):
        (qa, ta, ca), (qb, tb, cb) = pose[i], pose[j]
        o_want, _ = oracle_contact_list(objs[i][0], probes[i], ca, qa, ta, objs[j][0], probes[j], cb, qb, tb, 100 + i, 100 + j, i, j, resp)
        assert_contacts_equal(got[off[n] : off[n + 1]], o_want)
    # twice the same list: nothing kept between calls; and the empty list
    got2, off2 = many.mutual_voxel_object_contacts_many(many.mutual_queries(pairs))
    np.testing.assert_array_equal(off2, off)
    assert got2.tobytes() == got.tobytes()
    e, eo = many.mutual_voxel_object_contacts_many(many.mutual_queries([]))
    assert len(e) == 0 and eo.tolist() == [0]
    with pytest.raises(IvxError):  # too small a list is an error, not a truncation
        many.mutual_voxel_object_contacts_many(many.mutual_queries(pairs), capacity=8)
    with pytest.raises(IvxError):  # an object against itself
        bad = dict(pairs[0])
        bad["b"] = bad["a"]
        many.mutual_voxel_object_contacts_many(many.mutual_queries([bad]))
    for o, g in objs:
        g.close()


def test_probes_go_stale_with_the_mesh(ctx):
    from impact_amd.capi import IvxError

    A, GA = both(ctx, scenes.sphere_scene(20.0))
    B, GB = both(ctx, scenes.sphere_scene(20.0))
    GA.collision_probes_recompute()
    ident, zero = (0, 0, 0, 1), (0, 0, 0)
    with pytest.raises(IvxError):  # B has no probes yet
        GA.mutual_contacts(ident, zero, zero, GB, ident, zero, zero, 1, 2, 0, 1)
    GB.collision_probes_recompute()
    GA.mutual_contacts(ident, zero, zero, GB, ident, zero, zero, 1, 2, 0, 1)
    GA.mesh.recreate()  # a new mesh: A's probes are stale
    with pytest.raises(IvxError):
        GA.mutual_contacts(ident, zero, zero, GB, ident, zero, zero, 1, 2, 0, 1)
    GA.close()
    GB.close()


def world_to_object(body_q, body_p, model_centre):
    """transform_to_object_space of a voxel body whose body frame origin sits at `model_centre` of its model space:
    translate(model centre) . inverse(body pose), in f64 then rounded (both paths get the same f32 numbers)"""
    q = body_q.astype(np.float64)
    qi = np.array([-q[0], -q[1], -q[2], q[3]])
    tr = rot64(qi, -body_p.astype(np.float64)) + model_centre
    return qi.astype(f32), tr.astype(f32)


def test_voxel_body_dropped_on_a_voxel_body_steps_like_the_oracle(ctx):
    """the per-frame chain for two deformable bodies: mutual contacts from the current poses -> prepare -> solve -> integrate,
    70 steps; a small voxel sphere falls onto a big one, both dynamic; GPU contacts feed the GPU solver, oracle contacts the
    oracle's, rigid-body state within 1e-5 every step"""
    import physics_util as phu

    ext = 0.25
    A, GA = both(ctx, scenes.sphere_scene(24.0), ext)  # radius 6.0 world units, below
    B, GB = both(ctx, scenes.sphere_scene(10.0), ext)  # radius 2.5, falls
    pa, pb = probes_both(A, GA), probes_both(B, GB)
    ca, cb = A.center_of_mass(), B.center_of_mass()
    big = ol.uniform_sphere_body(6.0, 5.0, (0.0, 0.0, 0.0))
    small = ol.uniform_sphere_body(2.5, 2.0, (0.4, 6.0 + 2.5 + 0.04, -0.3), (0.0, -1.5, 0.0))
    dyn = np.array([big, small])
    dyn["total_force"][1] = (0.0, -9.81 * float(dyn["mass"][1]), 0.0)
    w, op = phu.make_pair(ctx, dyn)
    resp = (0.1, 0.6, 0.4)
    touched = 0
    for step in range(70):
        gd, od = w.bodies()[0], op.bodies()[0]
        qa_g, ta_g = world_to_object(gd["orientation"][0], gd["position"][0], ca.astype(np.float64))
        qb_g, tb_g = world_to_object(gd["orientation"][1], gd["position"][1], cb.astype(np.float64))
        qa_o, ta_o = world_to_object(od["orientation"][0], od["position"][0], ca.astype(np.float64))
        qb_o, tb_o = world_to_object(od["orientation"][1], od["position"][1], cb.astype(np.float64))
        got = GA.mutual_contacts(qa_g, ta_g, ca, GB, qb_g, tb_g, cb, 5, 6, 0, 1, resp)
        want, _ = oracle_contact_list(A, pa, ca, qa_o, ta_o, B, pb, cb, qb_o, tb_o, 5, 6, 0, 1, resp)
        assert len(got) == len(want), step
        touched += len(want) > 0
        w.perform_physics_step(got, 0.004)
        op.step(want, 0.004)
        phu.assert_bodies_close(w.bodies()[0], op.bodies()[0], what=f"step {step}: ")
    assert touched > 10
    w.close()
    GA.close()
    GB.close()
