import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
from impact_amd import capi, scenes
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject
ctx = Context(0)
gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(2.05), 0)
obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
obj.set_sdf_program(gen); obj.set_densities(np.ones(256, dtype=np.float32))
for _ in range(3): obj.step(capi.STAGE_ALL)
# run sample+derive only so the trace buffer (chunk moments) is not overwritten by the inertia stage
obj.step(capi.STAGE_SAMPLE | capi.STAGE_DERIVE)
L = capi.lib()
L.ivx_grid_device_ptr.restype = C.c_void_p
p = L.ivx_grid_device_ptr(obj.h, 6)
hip = C.CDLL('libamdhip64.so')
n = 7000
buf = np.zeros((n, 8), dtype=np.uint64)
hip.hipDeviceSynchronize()
rc = hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(p), buf.nbytes, 2)
assert rc == 0, rc
med = np.median(buf[:5000, 0].astype(np.float64))
ok = np.all(np.abs(buf[:, :8].astype(np.float64) - med) < 1e6, axis=1)
v = buf[ok]
t0 = v[:, 0].min()
d = (v[:, :6].astype(np.int64) - int(t0)) * 10e-3  # us (100 MHz clock)
print("items", len(v), "span us", d[:, 5].max())
print("mean durations us: loads", (d[:,1]-d[:,0]).mean(), "barrier1", (d[:,2]-d[:,1]).mean(), "counts+barrier2", (d[:,3]-d[:,2]).mean(), "flags", (d[:,4]-d[:,3]).mean(), "tail", (d[:,5]-d[:,4]).mean(), "total", (d[:,5]-d[:,0]).mean())
d2 = (v[:, 6:8].astype(np.int64) - int(t0)) * 10e-3
print("bbox", (d2[:,0]-d[:,2]).mean(), "face atomics", (d2[:,1]-d2[:,0]).mean(), "barrier2 wait", (d[:,3]-d2[:,1]).mean())
starts = np.sort(d[:, 0])
print("start time percentiles us", [round(float(np.percentile(starts, q)), 1) for q in (0, 10, 25, 50, 75, 90, 100)])
ends = np.sort(d[:, 5])
print("end percentiles us", [round(float(np.percentile(ends, q)), 1) for q in (0, 10, 25, 50, 75, 90, 100)])
# concurrency: items in flight at mid time
for t in (5, 10, 20, 30, 40):
    print("in flight at", t, "us:", int(np.sum((d[:,0] <= t) & (d[:,5] > t))))

