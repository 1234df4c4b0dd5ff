import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from impact_amd import capi, scenes
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject
ctx = Context(0)
gen = SDFVoxelGenerator(1.0, scenes.plates_scene(32), 0)
obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
obj.set_sdf_program(gen)
obj.set_densities(np.ones(256, dtype=np.float32))
obj.step(capi.STAGE_ALL)
lib = capi.lib(); lib.ivx_grid_device_ptr.restype = C.c_void_p
n = obj.n_chunks
hip = C.CDLL("libamdhip64.so")
lens = np.zeros(4 * n + 16, dtype=np.uint32); ops = np.zeros((n, 128, 2), dtype=np.uint32)
hip.hipDeviceSynchronize()
hip.hipMemcpy(lens.ctypes.data_as(C.c_void_p), C.c_void_p(lib.ivx_grid_device_ptr(obj.h, 7)), lens.nbytes, 2)
hip.hipMemcpy(ops.ctypes.data_as(C.c_void_p), C.c_void_p(lib.ivx_grid_device_ptr(obj.h, 8)), ops.nbytes, 2)
nodes = gen.nodes if hasattr(gen, "nodes") else None
for ch in [(16*32+16)*32+10, (5*32+7)*32+20]:
    ln = lens[ch]; print("chunk", ch, "len", ln)
    for i in range(ln):
        w, v = ops[ch, i]
        print(i, "op", w >> 28, "kind", (w >> 24) & 15, "node", w & 0xFFFFFF, "v", v, np.array([v], dtype=np.uint32).view(np.float32)[0])
print([a for a in dir(gen) if not a.startswith("_")])
