#!/usr/bin/env python3
"""Writes tests/golden/config5_golden.json: digests of BASELINE config 5's grid — the config-2 asteroid x4.2, 64^3 chunks = 1024^3 stored
voxels — from the CPU oracle (all cores; the per-chunk loops under OpenMP are checked bit-equal to the serial ones by
tests/test_oracle_parallel.py). Data only: sha-256 prefixes of the voxel planes, chunk-local labels, chunk records, mesh buffers; counts;
the ten f64 moments; region count; occupied ranges. tests/test_gpu_slabs.py checks the single-grid HIP step and the 8-slab decomposition
against it. Takes a few minutes and ~12 GB of memory.
usage: python tests/golden/make_golden_config5.py   (from the repository root; needs oracle/liboracle.so)"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as ol  # noqa: E402

OUT = os.path.join(HERE, "config5_golden.json")
FIELDS = ("kind", "gen_kind", "flags", "face_dist", "uniform_type", "region_count", "boundary_region_count")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()[:16]


def main():
    from impact_amd import scenes

    ol.build_oracle()
    threads = len(os.sched_getaffinity(0))
    t0 = time.perf_counter()
    o = ol.OracleObject.from_sdf_parallel(scenes.asteroid_scene(4.2), 1.0, 0, threads)
    assert tuple(o.chunk_counts) == (64, 64, 64)
    m = o.mesh_parallel(threads)
    _, m64 = o.inertia(np.ones(256, dtype=np.float32))
    n_regions, _ = o.region_labels()
    inf = o.info()
    sdf, typ, flg, lab, info = o.export_dense()
    rec = np.stack([info[f].astype(np.uint32) for f in FIELDS])
    d = {"_made_by": "tests/golden/make_golden_config5.py (CPU oracle, OpenMP over chunks)", "scene": "asteroid_scene(4.2)", "chunk_counts": [64, 64, 64],
         "voxel_sha": sha(sdf) + sha(typ) + sha(flg), "label_sha": sha(lab), "chunk_record_sha": sha(rec), "index_sha": sha(m.indices),
         "position_sha": sha(m.positions), "normal_sha": sha(m.normals), "index_material_sha": sha(m.index_materials),
         "triangles": int(m.indices.size // 3), "vertices": int(m.positions.shape[0]), "submeshes": int(m.submeshes.shape[0]), "regions": int(n_regions),
         "moments64": [float(x).hex() for x in m64],
         "occupied_chunk_ranges": [[int(a), int(b)] for a, b in inf["occupied_chunk_ranges"]],
         "occupied_voxel_ranges": [[int(a), int(b)] for a, b in inf["occupied_voxel_ranges"]],
         "non_empty_voxels": int(np.count_nonzero((flg & 1) == 0))}
    # per x-slab of 8 chunk planes: the voxel planes of the slab alone (what one of 8 ranks holds)
    per = 64 * 64 * 4096
    d["slab_voxel_sha"] = [sha(sdf[r * 8 * per:(r + 1) * 8 * per]) + sha(typ[r * 8 * per:(r + 1) * 8 * per]) + sha(flg[r * 8 * per:(r + 1) * 8 * per]) for r in range(8)]
    with open(OUT, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
    print("wrote", OUT, f"in {time.perf_counter() - t0:.0f} s:", d["triangles"], "triangles,", d["regions"], "region(s)")


if __name__ == "__main__":
    main()
