#!/usr/bin/env python3
"""Several ranks of the NATIVE slab protocol in ONE process, one thread (and one context = one HIP stream) per rank, over the shared-device
transport (ivx_comm_init_ipc). tests/test_gpu_slabs_ipc.py::test_config5_1024_as_8_ranks starts four of these with two ranks each: the GPU
box allows six processes on its card, so eight ranks are four processes of two — every neighbour exchange, the record gather, the sequence
counters and the error flags still run between eight independent ranks, across process boundaries at every second slab face.
Writes digests of what each rank ended with to <out>.json (the planes of an eighth of 1024^3 are not worth shipping).
usage: ipc_slab_worker8.py <first rank> <ranks here> <world> <shm name> <scene scale> <steps> <out>"""
import hashlib
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()[:16]


def main():
    first, here, world, name, scale, steps, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], float(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    from impact_amd import scenes
    from impact_amd.distributed import NativeComm, NativeSlabStepper, native_step
    from impact_amd.voxel import Context

    graph = scenes.asteroid_scene(scale)
    dens = np.ones(256, dtype=np.float32)
    results, errors = {}, []

    def run(rank):
        try:
            ctx = Context(0)
            comm = NativeComm(ctx, world, rank, ipc_name=name)
            st = NativeSlabStepper(ctx, comm, graph, dens, rank)
            for _ in range(steps):
                r = native_step([st])[0]
            sdf, typ, flg, _, _ = st.obj.download(labels=False, info=False)
            results[rank] = {"rank": rank, "x_range": [int(v) for v in st.x_range], "voxel_sha": sha16(sdf) + sha16(typ) + sha16(flg),
                             "region_count": int(r.region_count), "total_triangles": int(r.total_triangles), "vertices": int(r.mesh_counts[0]),
                             "indices": int(r.mesh_counts[1]), "vertex_offset": int(r.vertex_offset), "index_offset": int(r.index_offset),
                             "moments": [float(x).hex() for x in np.asarray(r.moments, dtype=np.float64)], "occupied": [int(x) for x in np.asarray(r.occupied).reshape(-1)],
                             "comm": comm.info()}
            st.close()
            comm.close()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(f"rank {rank}: {e!r}")

    threads = [threading.Thread(target=run, args=(first + i,)) for i in range(here)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        print("\n".join(errors))
        sys.exit(1)
    json.dump([results[first + i] for i in range(here)], open(out, "w"))


if __name__ == "__main__":
    main()
