#!/usr/bin/env python3
"""All N slabs of bench.py's N-rank workload in ONE process on one GPU (run_slabs_in_process: same protocol, device-to-device
copies instead of RCCL): wall time per step of the whole grid, and every slab's own stage-time sum — the slowest slab is what a
real N-GPU step waits for. usage: time_slabs_all.py N [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from impact_amd import scenes  # noqa: E402
from impact_amd.distributed import SlabStepper, run_slabs_in_process  # noqa: E402
from impact_amd.voxel import Context  # noqa: E402

n = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = Context(0)
dens = np.ones(256, dtype=np.float32)
graph = scenes.asteroid_scene(2.05 * n ** (1.0 / 3.0))
sts = [SlabStepper(ctx, graph, dens, r, n, torch) for r in range(n)]
for _ in range(3):
    res = run_slabs_in_process(sts)
ctx.synchronize()
t0 = time.perf_counter()
acc = np.zeros((n, len(res[0].stage_ms)))
for _ in range(steps):
    res = run_slabs_in_process(sts)
    for r in range(n):
        acc[r] += res[r].stage_ms
ctx.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
print(f"N={n} grid {sts[0].global_shape}: {wall:.3f} ms per step for all slabs on one GPU ({wall / n:.3f} ms per slab), regions {res[0].region_count}, "
      f"triangles {res[0].total_triangles}")
for r in range(n):
    print(f"  slab {r}: stage sum {acc[r].sum() / steps:.3f} ms, chunks {sts[r].obj.n_chunks}, sample {acc[r][0] / steps:.3f} derive {acc[r][1] / steps:.3f} "
          f"remesh {(acc[r][6] + acc[r][7] + acc[r][8]) / steps:.3f}")
for s in sts:
    s.close()
ctx.close()
