"""Shared helpers for the GPU-vs-oracle parity tests."""
from __future__ import annotations

import numpy as np

import oracle_lib as ol
from impact_amd.voxel import SDFVoxelGenerator, VoxelObject, VoxelObjectInertialPropertyManager, VoxelObjectMesh


def oracle_from_graph(graph, extent=1.0, vtype=0):
    o = ol.OracleObject.from_sdf(graph, extent, vtype)
    return o


def gpu_from_graph(ctx, graph, extent=1.0, vtype=0):
    gen = SDFVoxelGenerator(extent, graph, vtype)
    return VoxelObject.generate_without_derived_state(ctx, gen)


def assert_generated_equal(o: ol.OracleObject, g: VoxelObject):
    """voxel bytes and chunk classification right after generation"""
    o_sdf, o_typ, _, _, o_info = o.export_dense()
    g_sdf, g_typ, _, _, g_info = g.download(flags=False, labels=False)
    assert o.chunk_counts == g.chunk_counts
    np.testing.assert_array_equal(g_info["gen_kind"], o_info["gen_kind"])
    np.testing.assert_array_equal(g_sdf, o_sdf)
    np.testing.assert_array_equal(g_typ, o_typ)


def assert_derived_equal(o: ol.OracleObject, g: VoxelObject, check_regions=True):
    """flags, chunk kinds/flags/face distributions (+ local region counts)"""
    o_sdf, o_typ, o_flg, o_lab, o_info = o.export_dense()
    g_sdf, g_typ, g_flg, g_lab, g_info = g.download()
    np.testing.assert_array_equal(g_sdf, o_sdf)
    np.testing.assert_array_equal(g_typ, o_typ)
    np.testing.assert_array_equal(g_flg, o_flg)
    for f in ("kind", "gen_kind", "flags", "face_dist", "uniform_type"):
        np.testing.assert_array_equal(g_info[f], o_info[f], err_msg=f)
    if check_regions:
        for f in ("region_count", "boundary_region_count"):
            np.testing.assert_array_equal(g_info[f], o_info[f], err_msg=f)
        # chunk-local region labels: the raw u8 values of the reference (split_detection.rs:700-891), bit-exact
        np.testing.assert_array_equal(g_lab, o_lab)


def assert_mesh_equal(o: ol.OracleObject, g: VoxelObject, exact_normals=True):
    om = o.mesh()
    gm = VoxelObjectMesh.create(g)
    pos, nrm, idx, im, sub = gm.download()
    assert gm.n_vertices() == om.positions.shape[0]
    assert gm.n_indices() == om.indices.shape[0]
    assert gm.n_chunks() == om.submeshes.shape[0]
    np.testing.assert_array_equal(idx, om.indices)  # bit-exact triangle index buffer
    np.testing.assert_array_equal(pos.view(np.uint32), om.positions.view(np.uint32))  # bit-exact f32
    np.testing.assert_array_equal(im, om.index_materials)
    if exact_normals:
        np.testing.assert_array_equal(nrm.view(np.uint32), om.normals.view(np.uint32))
    else:
        np.testing.assert_allclose(nrm, om.normals, rtol=0, atol=2e-7)
    if len(sub):
        np.testing.assert_array_equal(sub["chunk_indices"], om.submeshes[:, 0:3])
        np.testing.assert_array_equal(sub["index_offset"], om.submeshes[:, 3])
        np.testing.assert_array_equal(sub["index_count"], om.submeshes[:, 4])
        np.testing.assert_array_equal(sub["is_obscured_from_direction"].reshape(-1, 8), om.submeshes[:, 5:13])
        np.testing.assert_array_equal(sub["vertex_offset"], om.submeshes[:, 13])
        np.testing.assert_array_equal(sub["vertex_count"], om.submeshes[:, 14])
    return gm


def assert_inertia_equal(o: ol.OracleObject, g: VoxelObject, densities=None, rtol=1e-5):
    d = np.ones(256, dtype=np.float32) if densities is None else densities
    _, o64 = o.inertia(d)
    mgr = VoxelObjectInertialPropertyManager.initialized_from(g, d)
    scale = np.maximum(np.abs(o64), 1e-300)
    assert np.all(np.abs(mgr.m64 - o64) <= rtol * scale + 1e-12), (mgr.m64, o64)
    return mgr


def assert_regions_equal(o: ol.OracleObject, g: VoxelObject):
    n, olab = o.region_labels()
    assert g.count_regions() == n
    glab = ol.tiled_to_dense(g.region_labels(), g.chunk_counts)
    np.testing.assert_array_equal(ol.canonicalize_labels(glab, 0xFFFFFFFF), ol.canonicalize_labels(olab, 0xFFFFFFFF))
    return n


def assert_edited_objects_equal(o: ol.OracleObject, g: VoxelObject, what="", densities=None, with_mesh=True):
    """after an operation that rewrites voxels (split, clip, absorb): voxel bytes, emptiness, types and flags of non-empty voxels,
    chunk-local labels, chunk records, regions, mesh and moments. Adjacency bits of EMPTY voxels are history artefacts in the
    reference (oracle/src/orc_split.cpp header) and are not compared."""
    o_sdf, o_typ, o_flg, o_lab, o_info = o.export_dense()
    g_sdf, g_typ, g_flg, g_lab, g_info = g.download()
    assert o.chunk_counts == g.chunk_counts, what
    np.testing.assert_array_equal(g_sdf, o_sdf, err_msg=what + "sdf")
    ne = (o_flg & 1) == 0
    np.testing.assert_array_equal((g_flg & 1) == 0, ne, err_msg=what + "emptiness")
    np.testing.assert_array_equal(g_typ[ne], o_typ[ne], err_msg=what + "types of non-empty voxels")
    np.testing.assert_array_equal(g_flg[ne], o_flg[ne], err_msg=what + "flags of non-empty voxels")
    np.testing.assert_array_equal(g_lab, o_lab, err_msg=what + "local labels")
    for f in ("kind", "flags", "face_dist", "region_count", "boundary_region_count"):
        np.testing.assert_array_equal(g_info[f], o_info[f], err_msg=what + f)
    n = assert_regions_equal(o, g)
    if with_mesh:  # (a full remesh of the HIP object: leave it out where the incremental remesh is what is being followed)
        assert_mesh_equal(o, g)
    assert_inertia_equal(o, g, densities)
    return n


def fuzz_seeds(default):
    """Seeds of a randomized parity test: the committed ones, or the range `IVX_FUZZ_SEEDS=a:b` names (a wider sweep run by hand
    on the GPU box; tools/fuzz_parity.sh)."""
    import os

    spec = os.environ.get("IVX_FUZZ_SEEDS")
    if not spec:
        return list(default)
    a, b = spec.split(":")
    return list(range(int(a), int(b)))


def fuzzing() -> bool:
    """True in a hand-run seed sweep: the 'this case exercised enough' assertions only hold for the committed seeds."""
    import os

    return bool(os.environ.get("IVX_FUZZ_SEEDS"))


def step_parity(o: ol.OracleObject, g: VoxelObject, res, densities=None, mesh=None) -> dict:
    """What one `ivx_voxel_step(STAGE_ALL)` left on the device against the oracle object built from the same SDF (derived state
    computed): sha-256 of the voxel bytes, chunk records, chunk-local labels, triangle index buffer, vertex positions; counts;
    relative error of the ten moments against the f64 oracle. Used by bench.py's `parity` fields and by the 512^3 parity test —
    the workload that is timed is the workload that is checked."""
    import hashlib

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()[:16]

    o_sdf, o_typ, o_flg, o_lab, o_info = o.export_dense()
    g_sdf, g_typ, g_flg, g_lab, g_info = g.download()
    om = mesh if mesh is not None else o.mesh()  # (`mesh`: the oracle mesh the caller already has, e.g. from the all-cores entry point)
    gm = VoxelObjectMesh(g)
    gm.counts = res["mesh"]
    pos, nrm, idx, im, sub = gm.download()
    d = np.ones(256, dtype=np.float32) if densities is None else densities
    _, o64 = o.inertia(d)
    g64 = np.asarray(res["moments"]["m64"], dtype=np.float64)
    n_regions, _ = o.region_labels()
    fields = ("kind", "gen_kind", "flags", "face_dist", "uniform_type", "region_count", "boundary_region_count")
    rec_o = np.stack([o_info[f].astype(np.uint32) for f in fields])
    rec_g = np.stack([g_info[f].astype(np.uint32) for f in fields])
    out = {
        "voxel_sha": [sha(g_sdf) + sha(g_typ) + sha(g_flg), sha(o_sdf) + sha(o_typ) + sha(o_flg)],
        "label_sha": [sha(g_lab), sha(o_lab)],
        "chunk_record_sha": [sha(rec_g), sha(rec_o)],
        "index_sha": [sha(idx), sha(om.indices)],
        "position_sha": [sha(pos), sha(om.positions)],
        "normal_sha": [sha(nrm), sha(om.normals)],
        "index_material_sha": [sha(im), sha(om.index_materials)],
        "triangles": [int(idx.size // 3), int(om.indices.size // 3)],
        "vertices": [int(pos.shape[0]), int(om.positions.shape[0])],
        "regions": [int(res["region_count"]), int(n_regions)],
        "moments_rel": float(np.max(np.abs(g64 - o64) / np.maximum(np.abs(o64), 1e-300))),
    }
    out["equal"] = bool(all(v[0] == v[1] for k, v in out.items() if isinstance(v, list)) and out["moments_rel"] <= 1e-5)
    return out
