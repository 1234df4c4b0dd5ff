"""Random contact graphs through the sequential-impulses solver (fixed seeds): irregular chains (bodies with one contact and with a dozen,
pairs given in either body order, manifolds of 1-4 contacts, a static plane), contact sets that persist, vanish and reappear between frames
in shuffled order. Every frame: body state within 1e-5 relative, ContactID order identical, warm-started impulses within 1e-4."""
import numpy as np
import pytest

import parity_util
import physics_util as pu
from impact_amd.capi import CONTACT_DTYPE, KINEMATIC_BIT
from impact_amd.physics import uniform_sphere_body
from test_gpu_physics import contact

pytestmark = pytest.mark.gpu
f32 = np.float32
BIT_REPORT_TURNED = []
BIT_REPORT = []  # (seed, words of body state that differed from the oracle) of test_random_graphs_on_several_workgroups


@pytest.mark.parametrize("seed", parity_util.fuzz_seeds([21, 22, 23, 24]))
def test_random_contact_graphs(ctx, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 60))
    pos = rng.uniform(-3.0, 3.0, (n, 3)).astype(f32)
    bodies = np.array([uniform_sphere_body(float(rng.uniform(0.3, 0.8)), float(rng.uniform(0.5, 3.0)), pos[i], rng.normal(0, 0.3, 3)) for i in range(n)])
    bodies["angular_momentum"] = rng.normal(0, 0.02, (n, 3)).astype(f32)
    w, o = pu.make_pair(ctx, bodies, pu.static_plane())
    n_pairs = int(rng.integers(1, 3 * n))
    pairs = []
    for k in range(n_pairs):
        a = int(rng.integers(0, n))
        if rng.random() < 0.15:
            b = KINEMATIC_BIT | 0
        else:
            b = int(rng.integers(0, n))
            if b == a:
                b = KINEMATIC_BIT | 0
        if rng.random() < 0.3 and not (b & KINEMATIC_BIT):
            a, b = b, a
        pairs.append((a, b, int(rng.integers(1, 5)), rng.random() < 0.5))
    alive = rng.random(n_pairs) < 0.7
    for frame in range(6):
        gd = w.bodies()[0]
        cs = []
        order = rng.permutation(n_pairs)
        for k in order:
            if not alive[k]:
                continue
            a, b, m, frictional = pairs[k]
            pa = gd[a]["position"].astype(np.float64)
            pb = np.array([pa[0], 0.0, pa[2]]) if (b & KINEMATIC_BIT) else gd[b]["position"].astype(np.float64)
            nrm = pa - pb
            nrm = nrm / np.linalg.norm(nrm) if np.linalg.norm(nrm) > 1e-6 else np.array([0.0, 1.0, 0.0])
            mid = 0.5 * (pa + pb)
            for j in range(m):
                jitter = rng.normal(0, 0.1, 3)
                geom = ((mid + jitter).astype(f32), nrm.astype(f32), f32(rng.uniform(-0.01, 0.05)))
                cs.append(contact(int(k) * 8 + j, a, b, geom, float(rng.uniform(0, 0.8)), 0.6 if frictional else 0.0, 0.4 if frictional else 0.0, first=(j == 0)))
        arr = np.array(cs) if cs else np.zeros(0, dtype=CONTACT_DTYPE)
        pu.step_both(w, o, arr, 0.004)
        pu.assert_bodies_close(w.bodies()[0], o.bodies()[0], what=f"seed {seed} frame {frame}: ")
        if len(arr):
            pu.compare_contact_state(w, o)
        flip = rng.random(n_pairs) < 0.25
        alive = np.where(flip, ~alive, alive)
    w.close()


@pytest.mark.parametrize("seed", parity_util.fuzz_seeds([31, 32, 33]))
def test_random_graphs_on_several_workgroups(ctx, seed):
    """the multi-workgroup solve (packed item records, grid barrier per level) on irregular input — manifolds of 1 to 9 contacts (chains
    shorter and longer than the four a packed record holds), a kinematic plane, pairs in either order, contact sets that change from
    frame to frame — against the single-workgroup kernel bit for bit, and that one against the oracle"""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(8, 60))
    pos = rng.uniform(-3.0, 3.0, (n, 3)).astype(f32)
    bodies = np.array([uniform_sphere_body(float(rng.uniform(0.3, 0.8)), float(rng.uniform(0.5, 3.0)), pos[i], rng.normal(0, 0.3, 3)) for i in range(n)])
    bodies["angular_momentum"] = rng.normal(0, 0.02, (n, 3)).astype(f32)
    plane = pu.static_plane()
    if seed % 2:  # a kinematic body that is turned, turns and moves (its orientation is not a fixed point of the re-normalisation
        # the reference applies to it in positional correction: the two-pass replay of DESIGN section 4 is what keeps these scenes exact)
        qk = np.random.default_rng(seed + 1_000_000).normal(size=4)
        qk = (qk / np.linalg.norm(qk)).astype(f32)
        if seed % 4 == 1:  # ... unless it IS a fixed point (re-normalised in f32, the solver's operation order, until it stays): exact in the
            # first frame without any replay; the body spins, so later frames start from a new orientation
            for _ in range(8):
                ln = np.sqrt(((qk[0] * qk[0] + qk[1] * qk[1]) + qk[2] * qk[2]) + qk[3] * qk[3], dtype=f32)
                qk = (qk / ln).astype(f32)
        plane["orientation"] = qk
        plane["angular_speed"] = 0.3
        plane["velocity"] = (0.05, 0.0, -0.02)
    w1, o = pu.make_pair(ctx, bodies, plane)
    wg, _ = pu.make_pair(ctx, bodies, plane)
    wc, _ = pu.make_pair(ctx, bodies, plane)  # (and the chain-stationary solve, k_solve_cs)
    groups = int(rng.integers(2, 5))
    w1.set_solver_groups(1)
    wg.set_solver_groups(groups)
    wc.set_solver_groups(255)
    n_pairs = int(rng.integers(4, 3 * n))
    pairs = []
    for k in range(n_pairs):
        a = int(rng.integers(0, n))
        b = (KINEMATIC_BIT | 0) if rng.random() < 0.2 else int(rng.integers(0, n))
        if b == a:
            b = KINEMATIC_BIT | 0
        if rng.random() < 0.3 and not (b & KINEMATIC_BIT):
            a, b = b, a
        pairs.append((a, b, int(rng.integers(1, 10)), rng.random() < 0.5))
    alive = rng.random(n_pairs) < 0.8
    differing = 0
    for frame in range(5):
        gd = w1.bodies()[0]
        cs = []
        for k in rng.permutation(n_pairs):
            if not alive[k]:
                continue
            a, b, m, frictional = pairs[k]
            pa = gd[a]["position"].astype(np.float64)
            pb = np.array([pa[0], 0.0, pa[2]]) if (b & KINEMATIC_BIT) else gd[b]["position"].astype(np.float64)
            nrm = pa - pb
            nrm = nrm / np.linalg.norm(nrm) if np.linalg.norm(nrm) > 1e-6 else np.array([0.0, 1.0, 0.0])
            mid = 0.5 * (pa + pb)
            for j in range(m):
                geom = ((mid + rng.normal(0, 0.1, 3)).astype(f32), nrm.astype(f32), f32(rng.uniform(-0.01, 0.05)))
                cs.append(contact(int(k) * 16 + j, a, b, geom, float(rng.uniform(0, 0.8)), 0.6 if frictional else 0.0, 0.4 if frictional else 0.0, first=(j == 0)))
        arr = np.array(cs) if cs else np.zeros(0, dtype=CONTACT_DTYPE)
        pu.step_both(w1, o, arr, 0.004)
        wg.perform_physics_step(arr, 0.004)
        wc.perform_physics_step(arr, 0.004)
        if len(arr):
            assert wg.solver_info()["workgroups"] == groups and w1.solver_info()["workgroups"] == 1
            assert wc.solver_info()["kernel"] == "chain_stationary" and w1.solver_info()["kernel"] == "one_workgroup"
        d1, dg, dc = w1.bodies()[0], wg.bodies()[0], wc.bodies()[0]
        for f in pu.STATE_FIELDS:
            np.testing.assert_array_equal(dg[f].view(np.uint32), d1[f].view(np.uint32), err_msg=f"seed {seed} frame {frame} {f}")
            np.testing.assert_array_equal(dc[f].view(np.uint32), d1[f].view(np.uint32), err_msg=f"seed {seed} frame {frame} {f} (chain-stationary)")
        if len(arr):
            np.testing.assert_array_equal(wg.contact_state()[1].view(np.uint32), w1.contact_state()[1].view(np.uint32))
            np.testing.assert_array_equal(wc.contact_state()[1].view(np.uint32), w1.contact_state()[1].view(np.uint32))
        if w1.n_kinematic:
            for f in ("position", "orientation"):
                np.testing.assert_array_equal(wc.bodies()[1][f].view(np.uint32), w1.bodies()[1][f].view(np.uint32), err_msg=f"seed {seed} frame {frame} kinematic {f}")
        od = o.bodies()[0]
        # (odd seeds: the turned kinematic body. The reference re-normalises a kinematic body's orientation at every positional correction
        # applied to it and uses the result for the corrections that follow; the schedule replays that — counts in a first pass, the
        # orientation each chain starts from in a second, DESIGN section 4 — and the state stays equal to the oracle's to the last bit)
        pu.assert_bodies_close(d1, od, what=f"seed {seed} frame {frame}: ")
        # The bar is 1e-5 relative; what is observed is more: every word of the state equal to the oracle's, frame after frame (the
        # orientation advance takes its sine and cosine from the double-precision functions rounded once, which is what libm's sinf /
        # cosf return). Counted, not required: a libm that rounds one argument differently must not fail the suite.
        differing += sum(int((d1[f].view(np.uint32) != od[f].view(np.uint32)).sum()) for f in pu.STATE_FIELDS)
        k1, ok = w1.bodies()[1], o.bodies()[1]  # the kinematic body is written back after the solve as well (solver.rs:571-602)
        for f in ("position", "orientation", "velocity", "angular_axis", "angular_speed"):
            np.testing.assert_allclose(k1[f], ok[f], rtol=1e-6, atol=1e-7, err_msg=f"seed {seed} frame {frame} kinematic {f}")
            differing += int((np.atleast_1d(k1[f]).view(np.uint32) != np.atleast_1d(ok[f]).view(np.uint32)).sum())
        alive = np.where(rng.random(n_pairs) < 0.25, ~alive, alive)
    w1.close()
    wg.close()
    wc.close()
    BIT_REPORT.append((seed, differing))
    if seed % 4 == 3:
        BIT_REPORT_TURNED.append((seed, differing))


def test_state_words_differing_from_the_oracle_are_reported():
    """(runs after the sweep above) how many words of body state differed from the oracle's, over all seeds and frames: 0 on the build
    and libm this was written on"""
    total = sum(d for _, d in BIT_REPORT)
    turned = sum(d for _, d in BIT_REPORT_TURNED)
    print("scenes with differing words (seed, words):", [(sd, d) for sd, d in BIT_REPORT if d][:12])
    print(f"random contact graphs: {len(BIT_REPORT)} seeds, {total} state words differing from the oracle "
          f"({turned} of them in the {len(BIT_REPORT_TURNED)} scenes whose kinematic orientation is not a fixed point of its re-normalisation)")
    assert len(BIT_REPORT) > 0
