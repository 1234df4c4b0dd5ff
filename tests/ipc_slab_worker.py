#!/usr/bin/env python3
"""One rank of tests/test_gpu_slabs_ipc.py: a fresh process that owns one x-slab of the scene and runs the NATIVE slab protocol
(ivx_slabs_step_enqueue / _collect) over the shared-device transport (ivx_comm_init_ipc). Writes what it ended with to <out>.npz.
usage: ipc_slab_worker.py <rank> <world> <shm name> <scene> <steps> <out>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, name, scene, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
    from impact_amd import capi, scenes
    from impact_amd.distributed import NativeComm, NativeSlabStepper, native_step
    from impact_amd.voxel import Context, VoxelObjectMesh

    graph = {"asteroid": scenes.asteroid_scene, "fracture": scenes.fracture_scene}[scene]()
    dens = np.linspace(0.5, 2.0, 256).astype(np.float32)
    ctx = Context(0)
    comm = NativeComm(ctx, world, rank, ipc_name=name)
    st = NativeSlabStepper(ctx, comm, graph, dens, rank)
    for _ in range(steps):  # (the second step starts from a dirty state: ghosts, labels, mesh buffers, sequence numbers)
        r = native_step([st])[0]
    sdf, typ, flg, lab, info = st.obj.download()
    m = VoxelObjectMesh(st.obj)
    m.counts = np.zeros((), dtype=capi.MESH_COUNTS_DTYPE)
    m.counts["n_vertices"], m.counts["n_indices"], m.counts["n_submeshes"] = r.mesh_counts
    pos, nrm, idx, im, sub = m.download()
    loc = st.obj.region_labels()
    np.savez(out, sdf=sdf, typ=typ, flg=flg, info=info, pos=pos, nrm=nrm, idx=idx, im=im, sub=sub, loc=loc, region_of_local=np.asarray(r.region_of_local),
             x_range=np.asarray(st.x_range), mesh_counts=np.asarray(r.mesh_counts), vertex_offset=r.vertex_offset, index_offset=r.index_offset,
             total_triangles=r.total_triangles, moments=np.asarray(r.moments), occupied=np.asarray(r.occupied), region_count=r.region_count,
             chunk_counts=np.asarray(st.global_chunk_counts))
    st.close()
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
