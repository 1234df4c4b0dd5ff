import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle_lib as ol, parity_util as pu
from impact_amd import scenes, many
from impact_amd.voxel import Context
ctx = Context(0)
graph = scenes.sphere_scene(20.0)
dens = np.linspace(0.5, 2.0, 256).astype(np.float32)
for mode in ("single_no_resident", "single_resident", "many_resident"):
    o = pu.oracle_from_graph(graph, 1.0); g = pu.gpu_from_graph(ctx, graph, 1.0)
    o.update_occupied_voxel_ranges(); o.compute_all_derived_state(); g.compute_all_derived_state(); g.update_occupied_voxel_ranges(); g.label_regions()
    if mode != "single_no_resident":
        g.set_densities(dens)
    c = np.array([24.0, 24.0, 44.0], dtype=np.float32)
    ro = o.absorb_sphere(c, 6.0, 4.0, dens)
    if mode == "many_resident":
        rg = many.absorb_sphere_many([g], [c], [6.0], [4.0], dens)[0]
    else:
        rg = g.absorb_sphere(c, 6.0, 4.0, dens)
    print(mode, "emptied", rg["emptied_voxels"], int(ro["emptied_by_type"].sum()), "mass gpu", rg["removed_moments"][0], "oracle", ro["removed64"][0])
    g.close()
