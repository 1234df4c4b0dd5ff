"""Digests of the voxel path's outputs, computed the same way from the oracle and from the HIP path, and the committed golden
vectors (tests/golden/voxel_golden.json, written by tests/golden/make_golden.py from the oracle — the reference itself is Rust
and cannot run in this image; the oracle is pinned by the reference's own known-answer tests in tests/test_oracle_*.py)."""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VOXEL_GOLDEN = os.path.join(GOLDEN_DIR, "voxel_golden.json")
PHYSICS_GOLDEN = os.path.join(GOLDEN_DIR, "physics_golden.json")


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:32]


def scenes_small():
    """name -> SDF graph: the BASELINE scenes at sizes the oracle finishes in about a second each"""
    from impact_amd import scenes

    return {
        "box_30": scenes.box_scene((30.0, 30.0, 30.0)),
        "sphere_r20": scenes.sphere_scene(20.0),
        "two_spheres_r12_sep30": scenes.two_spheres_scene(12.0, 30.0),
        "asteroid_x0.25": scenes.asteroid_scene(0.25),
        "fracture_x0.35": scenes.fracture_scene(0.35),
    }


def voxel_digest(chunk_counts, sdf, typ, flg, lab, info, mesh, moments64, region_count, canonical_labels, occupied):
    """mesh = (positions f32 [V,3], normals f32 [V,3], indices u32 [I], index_materials u8 [I,8], n_submeshes)"""
    pos, nrm, idx, im, ns = mesh
    return {
        "chunk_counts": [int(x) for x in chunk_counts],
        "sdf": sha(sdf), "type": sha(typ), "flags": sha(flg), "local_labels": sha(lab),
        "chunk_kind": sha(info["kind"]), "chunk_gen_kind": sha(info["gen_kind"]), "chunk_flags": sha(info["flags"]),
        "chunk_face_dist": sha(info["face_dist"]), "chunk_uniform_type": sha(info["uniform_type"]),
        "chunk_region_count": sha(info["region_count"]), "chunk_boundary_region_count": sha(info["boundary_region_count"]),
        "non_empty_voxels": int(np.count_nonzero((flg & 1) == 0)),
        "n_vertices": int(pos.shape[0]), "n_indices": int(idx.shape[0]), "n_submeshes": int(ns),
        "positions_bits": sha(pos.view(np.uint32)), "normals_bits": sha(nrm.view(np.uint32)),
        "indices": sha(idx), "index_materials": sha(im),
        "moments64": [float(x).hex() for x in moments64],
        "region_count": int(region_count), "canonical_labels": sha(canonical_labels),
        "occupied_chunk_ranges": [[int(a), int(b)] for a, b in occupied[0]],
        "occupied_voxel_ranges": [[int(a), int(b)] for a, b in occupied[1]],
    }


def oracle_voxel_digest(graph):
    import oracle_lib as ol

    o = ol.OracleObject.from_sdf(graph, 1.0, 0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    sdf, typ, flg, lab, info = o.export_dense()
    m = o.mesh()
    _, m64 = o.inertia(np.ones(256, dtype=np.float32))
    n, labels = o.region_labels()
    inf = o.info()
    return voxel_digest(o.chunk_counts, sdf, typ, flg, lab, info, (m.positions, m.normals, m.indices, m.index_materials, m.submeshes.shape[0]), m64,
                        n, ol.canonicalize_labels(labels, 0xFFFFFFFF), (inf["occupied_chunk_ranges"], inf["occupied_voxel_ranges"]))


def gpu_voxel_digest(ctx, graph):
    import oracle_lib as ol  # only for the label canonicalisation helpers (pure numpy)
    from impact_amd.voxel import SDFVoxelGenerator, VoxelObject, VoxelObjectInertialPropertyManager, VoxelObjectMesh

    g = VoxelObject.generate(ctx, SDFVoxelGenerator(1.0, graph, 0))
    sdf, typ, flg, lab, info = g.download()
    gm = VoxelObjectMesh.create(g)
    pos, nrm, idx, im, sub = gm.download()
    mgr = VoxelObjectInertialPropertyManager.initialized_from(g, np.ones(256, dtype=np.float32))
    n = g.count_regions()
    labels = ol.tiled_to_dense(g.region_labels(), g.chunk_counts)
    occ = (g.occupied_chunk_ranges, g.occupied_voxel_ranges)
    d = voxel_digest(g.chunk_counts, sdf, typ, flg, lab, info, (pos, nrm, idx, im, len(sub)), mgr.m64, n, ol.canonicalize_labels(labels, 0xFFFFFFFF), occ)
    g.close()
    return d


def load(path):
    with open(path) as f:
        return json.load(f)


MOMENT_RTOL = 1e-5  # the HIP path sums the same exact integer forms in another order


def assert_digest_equal(got, want, moments_exact):
    for k, v in want.items():
        if k == "moments64" and not moments_exact:
            a = np.array([float.fromhex(x) for x in got[k]]), np.array([float.fromhex(x) for x in v])
            assert np.all(np.abs(a[0] - a[1]) <= MOMENT_RTOL * np.maximum(np.abs(a[1]), 1e-300) + 1e-12), (k, got[k], v)
        else:
            assert got[k] == v, (k, got[k], v)
