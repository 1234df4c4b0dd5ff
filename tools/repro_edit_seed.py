#!/usr/bin/env python3
"""Developer tool: tests/test_gpu_edit_sequences.py's sequence of one seed, step by step, with what differs printed (touched / removed chunks,
the invalidated sets' difference with the chunks' coordinates and records) instead of asserted. usage: repro_edit_seed.py <seed>"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import oracle_lib as ol
import parity_util as pu
from impact_amd import scenes
from impact_amd.voxel import Context, VoxelObjectMesh

seed = int(sys.argv[1])
ctx = Context(0)
rng = np.random.default_rng(seed)
graph = scenes.asteroid_scene(0.3) if seed % 3 else scenes.box_scene((40.0, 26.0, 33.0))
o = pu.oracle_from_graph(graph, 1.0)
g = pu.gpu_from_graph(ctx, graph, 1.0)
o.update_occupied_voxel_ranges(); o.compute_all_derived_state()
g.compute_all_derived_state(); g.update_occupied_voxel_ranges(); g.label_regions()
om, gm = ol.OracleMeshHandle(o), VoxelObjectMesh.create(g)
occ = np.array(o.info()["occupied_voxel_ranges"], dtype=np.float64)
lo, hi = occ[:, 0], occ[:, 1]
cc = g.chunk_counts
for step in range(12):
    p = (lo + rng.uniform(-0.1, 1.1, 3) * (hi - lo)).astype(np.float32)
    r = float(np.float32(rng.uniform(2.0, 9.0)))
    if rng.uniform() < 0.5:
        kind = "sphere"
        ro, rg = o.absorb_sphere(p, r + 2.0, r), g.absorb_sphere(p, r + 2.0, r)
    else:
        v = (rng.normal(size=3) * rng.uniform(0.0, 25.0)).astype(np.float32)
        if step == 5:
            v[:] = 0.0
        kind = f"capsule v={v.tolist()}"
        ro, rg = o.absorb_capsule(p, v, r + 2.0, r), g.absorb_capsule(p, v, r + 2.0, r)
    same = np.array_equal(rg["invalidated"], ro["invalidated"])
    print(f"step {step}: {kind} p={p.tolist()} r={r}: touched {rg['touched_chunks']}/{ro['touched_chunks']} removed {rg['removed_chunks']}/{ro['removed_chunks']} "
          f"emptied {int(rg['emptied_by_type'].sum())}/{int(ro['emptied_by_type'].sum())} invalidated {int(rg['invalidated'].sum())}/{int(ro['invalidated'].sum())} {'same' if same else 'DIFFERENT'}")
    if not same:
        d = np.nonzero(rg["invalidated"] != ro["invalidated"])[0]
        info_g = g.download(sdf=False, types=False, flags=False, labels=False)[4]
        info_o = o.export_dense()[4]
        for c in d:
            ci, cj, ck = c // (cc[1] * cc[2]), (c // cc[2]) % cc[1], c % cc[2]
            print(f"   chunk {c} ({ci},{cj},{ck}): gpu {bool(rg['invalidated'][c])} oracle {bool(ro['invalidated'][c])}; kind gpu {info_g['kind'][c]} oracle {info_o['kind'][c]}")
    try:
        pu.assert_edited_objects_equal(o, g, with_mesh=False)
    except AssertionError as e:
        print("   objects differ:", str(e)[:300])
    om.sync(ro["invalidated"])
    gm.sync_with_voxel_object(ro["invalidated"])
