"""GPU parity of the x-slab domain decomposition (impact_amd/distributed.py): several slabs of one grid
live in ONE process on one GPU and run the same per-slab protocol the multi-GPU bench runs; halos move by
device copies instead of RCCL. Everything is compared with the oracle on the WHOLE grid: voxel bytes,
flags, chunk state, concatenated mesh (bit-exact index buffer after the per-slab vertex offset),
moments (1e-5 rel), occupied ranges and the connected-region partition."""
import numpy as np
import pytest

import oracle_lib as ol
from impact_amd import capi, scenes
from impact_amd.distributed import NativeComm, NativeSlabStepper, SlabStepper, native_step, run_slabs_in_process

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["native", "native_overlap", "native_copies", "python"])
def driver(request):
    """native: the protocol inside the library (`ivx_slabs_step_*` over an in-process communicator: the code the RCCL ranks run, the slabs
    reading their neighbours' send buffers in place); native_overlap: the same with the messages MOVED as under RCCL — copies on the
    communicator's own stream behind the packing, derive sweep and mesher count split around their arrival (`ivx_comm_set_local_copies(1)`);
    native_copies: copies on the context's stream, sweeps unsplit; python: the same phases driven from impact_amd/distributed.py (what the
    gloo protocol test shares)"""
    return request.param


def run_and_compare(ctx, graph, world, expect_regions=None, extent=1.0, driver="native"):
    import torch

    dens = np.linspace(0.5, 2.0, 256).astype(np.float32)
    o = ol.OracleObject.from_sdf(graph, extent, 0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    comm = None
    if driver.startswith("native"):
        comm = NativeComm(ctx, world, local=True)
        if driver != "native":
            comm.set_local_copies(1 if driver == "native_overlap" else 2)
        # (native_overlap also runs every slab's sampler pre-pass a step ahead, ivx_grid_set_sample_ahead: three steps, so the second and third
        # find theirs done)
        steppers = [NativeSlabStepper(ctx, comm, graph, dens, r, extent, sample_ahead=driver == "native_overlap") for r in range(world)]
    else:
        steppers = [SlabStepper(ctx, graph, dens, r, world, torch, extent) for r in range(world)]
    try:
        for _ in range(3 if driver == "native_overlap" else 2):  # again: a later pass starts from a dirty state (ghosts, labels, mesh buffers, receive buffers)
            results = native_step(steppers) if driver.startswith("native") else run_slabs_in_process(steppers)
        cc = o.chunk_counts
        assert steppers[0].global_chunk_counts == cc
        o_sdf, o_typ, o_flg, o_lab, o_info = o.export_dense()
        per_chunk = cc[1] * cc[2]
        # voxel planes + chunk state, slab by slab
        for s in steppers:
            x0, x1 = s.x_range
            sl = slice(x0 * per_chunk * 4096, x1 * per_chunk * 4096)
            g_sdf, g_typ, g_flg, g_lab, g_info = s.obj.download()
            np.testing.assert_array_equal(g_sdf, o_sdf[sl])
            np.testing.assert_array_equal(g_typ, o_typ[sl])
            np.testing.assert_array_equal(g_flg, o_flg[sl])
            oi = o_info[x0 * per_chunk:x1 * per_chunk]
            for f in ("kind", "gen_kind", "flags", "face_dist", "uniform_type", "region_count", "boundary_region_count"):
                np.testing.assert_array_equal(g_info[f], oi[f], err_msg=f"{f} slab {s.rank}")
        # mesh: concatenation of the slab meshes in rank order
        om = o.mesh()
        pos, nrm, idx, im, sub = [], [], [], [], []
        from impact_amd.voxel import VoxelObjectMesh

        for s, r in zip(steppers, results):
            m = VoxelObjectMesh(s.obj)
            m.counts = np.zeros((), dtype=capi.MESH_COUNTS_DTYPE)
            m.counts["n_vertices"], m.counts["n_indices"], m.counts["n_submeshes"] = r.mesh_counts
            p, n, i, mat, sm = m.download()
            pos.append(p)
            nrm.append(n)
            idx.append(i + np.uint32(r.vertex_offset))
            im.append(mat)
            sm = sm.copy()
            sm["index_offset"] += r.index_offset
            sm["vertex_offset"] += r.vertex_offset
            sub.append(sm)
        pos, nrm, idx, im, sub = np.concatenate(pos), np.concatenate(nrm), np.concatenate(idx), np.concatenate(im), np.concatenate(sub)
        np.testing.assert_array_equal(idx, om.indices)
        np.testing.assert_array_equal(pos.view(np.uint32), om.positions.view(np.uint32))
        np.testing.assert_array_equal(nrm.view(np.uint32), om.normals.view(np.uint32))
        np.testing.assert_array_equal(im, om.index_materials)
        np.testing.assert_array_equal(sub["chunk_indices"], om.submeshes[:, 0:3])
        np.testing.assert_array_equal(sub["index_offset"], om.submeshes[:, 3])
        np.testing.assert_array_equal(sub["vertex_offset"], om.submeshes[:, 13])
        assert results[0].total_triangles == om.indices.size // 3
        # moments, occupied ranges
        _, o64 = o.inertia(dens)
        for r in results:
            np.testing.assert_allclose(r.moments, o64, rtol=1e-5)
            np.testing.assert_array_equal(r.moments, results[0].moments)  # identical on every rank
        info = o.info()
        occ = results[0].occupied
        assert [(int(occ[2 * d]), int(occ[2 * d + 1])) for d in range(3)] == info["occupied_chunk_ranges"]
        assert [(int(occ[6 + 2 * d]), int(occ[7 + 2 * d])) for d in range(3)] == info["occupied_voxel_ranges"]
        # connected regions: same count and same partition of the voxels
        n_o, olab = o.region_labels()
        assert all(r.region_count == n_o for r in results)
        if expect_regions is not None:
            assert n_o == expect_regions
        glab = []
        for s, r in zip(steppers, results):
            loc = s.obj.region_labels()
            out = np.full(loc.shape, 0xFFFFFFFF, dtype=np.uint32)
            m = loc != 0xFFFFFFFF
            out[m] = r.region_of_local[loc[m]]
            glab.append(out)
        glab = ol.tiled_to_dense(np.concatenate(glab), cc)
        np.testing.assert_array_equal(ol.canonicalize_labels(glab, 0xFFFFFFFF), ol.canonicalize_labels(olab, 0xFFFFFFFF))
    finally:
        for s in steppers:
            s.close()
        if comm is not None:
            comm.close()


@pytest.mark.parametrize("world", [2, 4])
def test_asteroid_256_in_slabs(ctx, world, driver):
    """BASELINE config 2 body, split in 2 and 4 x-slabs (config 5's decomposition at 1/4 scale)"""
    run_and_compare(ctx, scenes.asteroid_scene(), world, expect_regions=1, driver=driver)


def test_fracture_256_in_3_slabs(ctx, driver):
    """config 3: 8 octants; an uneven 3-way split puts a slab boundary inside four of them"""
    run_and_compare(ctx, scenes.fracture_scene(), 3, expect_regions=8, driver=driver)


def test_asteroid_row_in_slabs(ctx, driver):
    """the weak-scaling bench workload at small scale: one body per slab joined by a bar"""
    run_and_compare(ctx, scenes.asteroid_row_scene(3, 0.25), 3, expect_regions=1, driver=driver)


def test_two_spheres_cut_between(ctx, driver):
    """a slab boundary in the gap between two bodies: empty ghost planes, two regions"""
    run_and_compare(ctx, scenes.two_spheres_scene(25.0, 60.0), 2, expect_regions=2, extent=0.5, driver=driver)


@pytest.mark.parametrize("overlap", [False, True])
def test_headline_512_in_8_slabs(ctx, overlap):
    """the strong-scaling configuration of the metric — the 512^3 asteroid in 8 x-slabs of 4 chunk planes — through the native driver,
    all eight slabs on this one GPU: global results against the single-grid step of the same scene. `overlap`: the messages moved on the
    communicator's stream and the sweeps split around their arrival, as under RCCL (three steps: the receive buffers are reused)"""
    from impact_amd.voxel import SDFVoxelGenerator, VoxelObject

    graph = scenes.asteroid_scene(2.05)
    dens = np.ones(256, dtype=np.float32)
    gen = SDFVoxelGenerator(1.0, graph, 0)
    whole = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    whole.set_sdf_program(gen)
    whole.set_densities(dens)
    ref = whole.step(capi.STAGE_ALL)
    comm = NativeComm(ctx, 8, local=True)
    if overlap:
        comm.set_local_copies(1)
    steppers = [NativeSlabStepper(ctx, comm, graph, dens, r, sample_ahead=overlap) for r in range(8)]  # (overlap: the slabs' pre-passes a step ahead too)
    try:
        for _ in range(3 if overlap else 1):
            results = native_step(steppers)
        assert results[0].region_count == int(ref["region_count"]) == 1
        assert results[0].total_triangles == int(ref["mesh"]["n_indices"]) // 3
        assert sum(r.mesh_counts[0] for r in results) == int(ref["mesh"]["n_vertices"])
        np.testing.assert_allclose(results[0].moments, np.asarray(ref["moments"]["m64"]), rtol=1e-12)
        np.testing.assert_array_equal(results[0].occupied, np.asarray(ref["occupied"]))
        # every slab's voxel bytes equal the whole grid's
        w_sdf, w_typ, w_flg, _, _ = whole.download(labels=False, info=False)
        per = gen.chunk_counts()[1] * gen.chunk_counts()[2] * 4096
        for s in steppers:
            x0, x1 = s.x_range
            g_sdf, g_typ, g_flg, _, _ = s.obj.download(labels=False, info=False)
            np.testing.assert_array_equal(g_sdf, w_sdf[x0 * per:x1 * per])
            np.testing.assert_array_equal(g_flg, w_flg[x0 * per:x1 * per])
    finally:
        for s in steppers:
            s.close()
        comm.close()
        whole.close()


def _sha16(a):
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()[:16]


def test_config5_1024_in_8_slabs(ctx):
    """BASELINE config 5 — the 1024^3 grid (config-2 asteroid x4.2: 64^3 chunks) — as one grid and domain-decomposed in 8 x-slabs of 8 chunk
    planes through the native driver (all eight slabs on this one GPU), BOTH against the oracle's digests of that grid
    (tests/golden/config5_golden.json, written by tests/golden/make_golden_config5.py): voxel planes, chunk-local labels, chunk records, every mesh
    buffer, counts, moments, regions, occupied ranges for the single grid; every slab's voxel planes and the protocol's global results for the slabs"""
    import json
    import os

    from impact_amd.voxel import SDFVoxelGenerator, VoxelObject, VoxelObjectMesh

    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config5_golden.json")))
    m64 = np.array([float.fromhex(x) for x in gold["moments64"]])
    graph = scenes.asteroid_scene(4.2)
    dens = np.ones(256, dtype=np.float32)
    gen = SDFVoxelGenerator(1.0, graph, 0)
    assert tuple(gen.chunk_counts()) == (64, 64, 64) == tuple(gold["chunk_counts"])
    whole = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    whole.set_sdf_program(gen)
    whole.set_densities(dens)
    ref = whole.step(capi.STAGE_ALL)
    try:
        # ---- the single grid against the oracle
        w_sdf, w_typ, w_flg, w_lab, w_info = whole.download()
        assert _sha16(w_sdf) + _sha16(w_typ) + _sha16(w_flg) == gold["voxel_sha"]
        assert _sha16(w_lab) == gold["label_sha"]
        fields = ("kind", "gen_kind", "flags", "face_dist", "uniform_type", "region_count", "boundary_region_count")
        assert _sha16(np.stack([w_info[f].astype(np.uint32) for f in fields])) == gold["chunk_record_sha"]
        assert int(np.count_nonzero((w_flg & 1) == 0)) == gold["non_empty_voxels"]
        del w_lab, w_info
        gm = VoxelObjectMesh(whole)
        gm.counts = ref["mesh"]
        pos, nrm, idx, im, sub = gm.download()
        assert (idx.size // 3, pos.shape[0], sub.shape[0]) == (gold["triangles"], gold["vertices"], gold["submeshes"])
        assert _sha16(idx) == gold["index_sha"] and _sha16(im) == gold["index_material_sha"]
        assert _sha16(pos) == gold["position_sha"] and _sha16(nrm) == gold["normal_sha"]
        del pos, nrm, idx, im, sub
        assert int(ref["region_count"]) == gold["regions"]
        g64 = np.asarray(ref["moments"]["m64"], dtype=np.float64)
        assert float(np.max(np.abs(g64 - m64) / np.maximum(np.abs(m64), 1e-300))) <= 1e-5
        occ = np.asarray(ref["occupied"]).reshape(-1)
        want_occ = np.array([x for r in gold["occupied_chunk_ranges"] for x in r] + [x for r in gold["occupied_voxel_ranges"] for x in r])  # (lo, hi) per axis
        np.testing.assert_array_equal(occ, want_occ)
        # ---- the eight slabs against the oracle (and against the single grid, voxel byte for voxel byte)
        comm = NativeComm(ctx, 8, local=True)
        steppers = [NativeSlabStepper(ctx, comm, graph, dens, r) for r in range(8)]
        try:
            results = native_step(steppers)
            assert results[0].region_count == gold["regions"]
            assert results[0].total_triangles == gold["triangles"]
            assert sum(r.mesh_counts[0] for r in results) == gold["vertices"]
            rm = np.asarray(results[0].moments, dtype=np.float64)
            assert float(np.max(np.abs(rm - m64) / np.maximum(np.abs(m64), 1e-300))) <= 1e-5
            np.testing.assert_allclose(results[0].moments, g64, rtol=1e-12)
            np.testing.assert_array_equal(np.asarray(results[0].occupied).reshape(-1), want_occ)
            per = gen.chunk_counts()[1] * gen.chunk_counts()[2] * 4096
            for r, s in enumerate(steppers):
                x0, x1 = s.x_range
                g_sdf, g_typ, g_flg, _, _ = s.obj.download(labels=False, info=False)
                assert _sha16(g_sdf) + _sha16(g_typ) + _sha16(g_flg) == gold["slab_voxel_sha"][r], f"slab {r}"
                assert np.array_equal(g_sdf, w_sdf[x0 * per:x1 * per]) and np.array_equal(g_flg, w_flg[x0 * per:x1 * per]), f"slab {x0}:{x1}"
                del g_sdf, g_typ, g_flg
        finally:
            for s in steppers:
                s.close()
            comm.close()
    finally:
        whole.close()


def test_rccl_binding_selftest(ctx):
    """the run-time binding to librccl on the one GPU there is: unique id, a one-rank communicator on the context's device, the grouped
    send / receive pair of the neighbour exchange (to the rank itself) and the record all-gather on the library's stream, bytes checked"""
    from impact_amd.capi import check, lib

    check(lib().ivx_comm_selftest(ctx.h))


@pytest.mark.parametrize("seed", __import__("parity_util").fuzz_seeds([41, 42, 43, 44]))
def test_random_sdf_program_in_slabs(ctx, seed):
    """random SDF programs (tests/test_gpu_random_sdf.py's trees) cut into 2-4 x-slabs wherever the cuts happen to fall — through bodies,
    through gaps, through smooth blends — against the oracle's whole grid: voxel bytes, chunk state, the concatenated mesh, moments,
    occupied ranges and the partition into regions"""
    from impact_amd.sdf_graph import SDFGraph
    from impact_amd.voxel import SDFVoxelGenerator
    from test_gpu_random_sdf import random_tree

    rng = np.random.default_rng(seed)
    g = SDFGraph()
    random_tree(g, rng, int(rng.integers(1, 4)))
    extent = [1.0, 0.5][seed % 2]
    cx = SDFVoxelGenerator(extent, g, 0).chunk_counts()[0]
    if cx < 2:  # degenerate or one chunk plane thick: nothing to decompose
        return
    world = int(min(cx, rng.integers(2, 5)))
    # (every third seed with the messages moved on the communicator's stream and the sweeps split around their arrival)
    run_and_compare(ctx, g, world, extent=extent, driver="native_overlap" if seed % 3 == 0 else "native")
