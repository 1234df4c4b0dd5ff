import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle_lib as ol, parity_util as pu
from impact_amd import scenes
from impact_amd.voxel import Context, VoxelObjectMesh
ctx = Context(0)
graph = scenes.asteroid_scene(0.5)
o = pu.oracle_from_graph(graph, 1.0); g = pu.gpu_from_graph(ctx, graph, 1.0)
o.update_occupied_voxel_ranges(); o.compute_all_derived_state(); g.compute_all_derived_state(); g.update_occupied_voxel_ranges(); g.label_regions()
ctr = np.array([0.5 * (a + b) for a, b in o.info()["occupied_voxel_ranges"]], dtype=np.float32)
for step, (d, r) in enumerate([((0.0, 0.0, 48.0), 14.0), ((30.0, 5.0, 30.0), 9.0), ((0.0, 0.0, 40.0), 22.0), ((-50.0, 0.0, 0.0), 30.0)]):
    c = ctr + np.asarray(d, dtype=np.float32)
    ro = o.absorb_sphere(c, r + 2.0, r)
    mode = sys.argv[1] if len(sys.argv) > 1 else "split"
    if mode == "split":
        g.absorb_sphere_enqueue(c, r + 2.0, r); rg = g.absorb_collect()
    else:
        rg = g.absorb_sphere(c, r + 2.0, r)
    o_sdf, o_typ, o_flg, o_lab, o_info = o.export_dense()
    g_sdf, g_typ, g_flg, g_lab, g_info = g.download()
    bad = np.nonzero(g_sdf != o_sdf)[0]
    print("step", step, "touched", rg["touched_chunks"], ro["touched_chunks"], "removed", rg["removed_chunks"], ro["removed_chunks"], "bad voxels", bad.size,
          "occ oracle", o.info()["occupied_voxel_ranges"])
    if bad.size:
        ch = np.unique(bad // 4096)
        cc = g.chunk_counts
        for c_ in ch[:12]:
            print("  chunk", c_, (c_ // (cc[1]*cc[2]), (c_ // cc[2]) % cc[1], c_ % cc[2]), "gpu kind/gen", g_info["kind"][c_], g_info["gen_kind"][c_], "oracle kind", o_info["kind"][c_],
                  "gpu sdf sample", g_sdf[c_*4096:c_*4096+4], "oracle", o_sdf[c_*4096:c_*4096+4], "n bad", int((bad // 4096 == c_).sum()))
        break
