import sys, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np
from impact_amd import capi, scenes
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject
ctx = Context(0)
gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(2.05), 0)
obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
obj.set_sdf_program(gen); obj.set_densities(np.ones(256, dtype=np.float32))
ms = []
for s in range(6):
    res = obj.step(capi.STAGE_ALL)
    ms.append(float(res["moments"]["m64"][0]))
print("mass per step", ms, "expected 35141832.0", os.uname().nodename)
