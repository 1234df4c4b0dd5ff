"""SURVEY §8f-4, the mesh hand-off: the buffers the renderer consumes (gpu_resource.rs:498-530, 729-907; mesh.rs:94-123) exported as handles another
process can import (ivx_mesh_export: hipIpc handle + dma-buf descriptor). A second, fresh process — standing in for the renderer — opens the
handles and must read exactly the bytes ivx_mesh_download returns, after a full remesh and again after an edit + incremental sync."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from impact_amd import scenes
from impact_amd.voxel import SDFVoxelGenerator, VoxelObject, VoxelObjectMesh

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = VoxelObjectMesh.MESH_BUFFERS


def import_in_another_process(mesh, tmp, tag):
    exp = {n: mesh.export(n) for n in NAMES}
    req = {}
    for n in NAMES:
        req[n + "_handle"] = np.frombuffer(exp[n]["ipc_handle"], dtype=np.uint8)
        req[n + "_bytes"] = np.int64(exp[n]["bytes"])
    rq, out = os.path.join(tmp, f"req_{tag}.npz"), os.path.join(tmp, f"out_{tag}.npz")
    np.savez(rq, **req)
    r = subprocess.run([sys.executable, os.path.join(HERE, "mesh_import_worker.py"), rq, out], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return exp, np.load(out)


def test_mesh_buffers_imported_by_another_process(ctx):
    g = VoxelObject.generate(ctx, SDFVoxelGenerator(1.0, scenes.sphere_scene(30.0), 0))
    try:
        mesh = VoxelObjectMesh.create(g)
        with tempfile.TemporaryDirectory() as tmp:
            exp, got = import_in_another_process(mesh, tmp, "full")
            want = dict(zip(NAMES, mesh.download()))
            for n in NAMES:
                assert exp[n]["bytes"] == want[n].nbytes and exp[n]["capacity_bytes"] >= exp[n]["bytes"]
                np.testing.assert_array_equal(got[n], np.ascontiguousarray(want[n]).view(np.uint8).reshape(-1), err_msg=n)
                fd = exp[n]["dmabuf_fd"]
                if fd >= 0:  # a dma-buf descriptor where the runtime makes one: it is one, and it covers the buffer
                    assert "dmabuf" in os.readlink(f"/proc/self/fd/{fd}")
                    assert os.fstat(fd).st_size == 0 or os.fstat(fd).st_size >= exp[n]["bytes"]
                    os.close(fd)
            gen0 = mesh.generation()
            assert all(exp[n]["generation"] == gen0 for n in NAMES)
            # an edit and the incremental remesh: live ranges move inside the (possibly regrown) buffers; the renderer re-imports when the
            # generation moved and reads the whole capacity
            res = g.absorb_sphere((3.0, -2.0, 20.0), 9.0, 8.0)
            mesh.sync_with_voxel_object(res["invalidated"])
            exp2, got2 = import_in_another_process(mesh, tmp, "sync")
            if mesh.generation() == gen0:
                assert all(exp2[n]["ipc_handle"] == exp[n]["ipc_handle"] for n in NAMES)  # same allocations: the old handles stay good
            pos, nrm, idx, im, sub = mesh.download()
            for n, w in zip(NAMES, (pos, nrm, idx, im, sub)):
                wb = np.ascontiguousarray(w).view(np.uint8).reshape(-1)
                np.testing.assert_array_equal(got2[n][:wb.size], wb, err_msg=n + " after sync")
                if exp2[n]["dmabuf_fd"] >= 0:
                    os.close(exp2[n]["dmabuf_fd"])
    finally:
        g.close()
