import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib as ol, parity_util as pu
from impact_amd import scenes
from impact_amd.voxel import Context
ctx = Context(0)
o = pu.oracle_from_graph(scenes.sphere_scene(60.0)); g = pu.gpu_from_graph(ctx, scenes.sphere_scene(60.0))
o.update_occupied_voxel_ranges(); o.compute_all_derived_state(); g.compute_all_derived_state(); g.update_occupied_voxel_ranges(); g.label_regions()
c = np.array([0.5*(a+b) for a,b in o.info()["occupied_voxel_ranges"]], dtype=np.float32) + np.array([0.25,-0.5,0.75],np.float32)
o.absorb_sphere(c, 19.0, 17.0); g.absorb_sphere(c, 19.0, 17.0)
o_sdf,o_typ,o_flg,o_lab,o_info = o.export_dense(); g_sdf,g_typ,g_flg,g_lab,g_info = g.download()
ne = (o_flg & 1) == 0
bad = np.nonzero(ne & (o_flg != g_flg))[0]
print("mismatches", len(bad))
ch = bad >> 12
u, cnt = np.unique(ch, return_counts=True)
cc = o.chunk_counts
for c_, n_ in list(zip(u, cnt))[:12]:
    print("chunk", c_, (c_//(cc[1]*cc[2]), (c_//cc[2])%cc[1], c_%cc[2]), "n", n_, "o kind/gen", o_info["kind"][c_], o_info["gen_kind"][c_], "g kind/gen", g_info["kind"][c_], g_info["gen_kind"][c_],
          "xor bits", np.unique(o_flg[bad[ch==c_]] ^ g_flg[bad[ch==c_]]))
i = bad[0]; print("first: idx in chunk", i & 4095, ((i&4095)>>8, ((i&4095)>>4)&15, i&15), "o", o_flg[i], "g", g_flg[i])
