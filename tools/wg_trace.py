#!/usr/bin/env python3
"""Workgroup timelines of a list-driven kernel (developer tool; needs the library built with `make -C impact_amd/csrc TRACE=1`).
Kernels compiled with IVX_WG_TRACE store up to six 100 MHz wall-clock stamps of wave 0 per list entry (IVX_T in
ivx_internal.hpp) into the grid's per-chunk moment buffer; this script runs the 512^3 bench workload up to the stage asked for,
reads the stamps back and prints the mean time between consecutive probes, start/end percentiles and the number of entries in
flight. Only ONE traced kernel may run in the traced step (they share the buffer), so choose the stage whose kernel has probes.
usage: wg_trace.py {prepass|sample|derive|remesh} [scale|dense]   (sample: k_sdf_eval, derive: k_derive with the region pass, remesh: k_sn_emit)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from impact_amd import capi, scenes  # noqa: E402
from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject  # noqa: E402

STAGES = {"prepass": capi.STAGE_SAMPLE, "sample": capi.STAGE_SAMPLE, "derive": capi.STAGE_DERIVE | capi.STAGE_REGIONS, "remesh": capi.STAGE_REMESH}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "remesh"
    dense = len(sys.argv) > 2 and sys.argv[2] == "dense"  # the all-surface workload (32 perforated plates) instead of the asteroid
    scale = float(sys.argv[2]) if len(sys.argv) > 2 and not dense else 2.05
    ctx = Context(0)
    gen = SDFVoxelGenerator(1.0, scenes.plates_scene(32) if dense else scenes.asteroid_scene(scale), 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    for _ in range(3):
        obj.step(capi.STAGE_ALL & ~capi.STAGE_INERTIA)
    obj.step(STAGES[which])
    lib = capi.lib()
    lib.ivx_grid_device_ptr.restype = C.c_void_p
    p = lib.ivx_grid_device_ptr(obj.h, 6)
    if not p:
        raise SystemExit("the library was not built with TRACE=1")
    hip = C.CDLL("libamdhip64.so")
    n = min(obj.n_chunks, 40000)
    buf = np.zeros((n, 8), dtype=np.uint64)
    hip.hipDeviceSynchronize()
    assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(p), buf.nbytes, 2) == 0
    if which == "prepass":  # entries = super-blocks of 4^3 chunks
        n = int(np.prod([(c + 3) // 4 for c in obj.chunk_counts]))
        buf = buf[:n]
    med = np.median(buf[: max(n // 4, 1), 0].astype(np.float64))
    ok = np.all(np.abs(buf[:, :6].astype(np.float64) - med) < 1e6, axis=1)  # slots of entries that were not written hold old data
    v = buf[ok]
    # the buffer is shared by every traced kernel: entries beyond this kernel's list keep an earlier kernel's stamps. Keep the LAST
    # cluster of start times (the traced kernel ran last), split at the widest gap of more than 30 us.
    order = np.argsort(v[:, 0])
    v = v[order]
    gaps = np.diff(v[:, 0].astype(np.int64))
    if len(gaps) and gaps.max() > 3000:
        v = v[int(np.argmax(gaps)) + 1:]
    t0 = int(v[:, 0].min())
    d = (v[:, :6].astype(np.int64) - t0) * 0.01  # us
    print(f"entries {len(v)}, span {d[:, 5].max():.1f} us")
    print("mean us between probes:", " ".join(f"{(d[:, i + 1] - d[:, i]).mean():.2f}" for i in range(5)), f"| total {(d[:, 5] - d[:, 0]).mean():.2f}")
    if v[:, 6].max() > 0:
        e = (v[:, 6:8].astype(np.int64) - t0) * 0.01
        print("extra probes after probe 0 (us): 6:", round(float((e[:, 0] - d[:, 0]).mean()), 2), " 7:", round(float((e[:, 1] - d[:, 0]).mean()), 2))
    print("start percentiles us", [round(float(np.percentile(d[:, 0], q)), 1) for q in (0, 10, 25, 50, 75, 90, 100)])
    print("end percentiles us", [round(float(np.percentile(d[:, 5], q)), 1) for q in (0, 10, 25, 50, 75, 90, 100)])
    dur = d[:, 5] - d[:, 0]
    print("entry duration percentiles us", [round(float(np.percentile(dur, q)), 1) for q in (0, 10, 50, 90, 100)])
    for t in (5, 10, 20, 40, 60, 80) + ((200, 400, 600) if dense else ()):
        print(f"in flight at {t} us: {int(np.sum((d[:, 0] <= t) & (d[:, 5] > t)))}")


if __name__ == "__main__":
    main()
