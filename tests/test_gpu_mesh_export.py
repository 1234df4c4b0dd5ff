"""SURVEY §8f-4, the mesh hand-off: the buffers the renderer consumes (gpu_resource.rs:498-530, 729-907; mesh.rs:94-123) exported as handles another
process can import (ivx_mesh_export: hipIpc handle + dma-buf descriptor). A second, fresh process — standing in for the renderer — opens the
handles and must read exactly the bytes ivx_mesh_download returns, after a full remesh and again after an edit + incremental sync."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from impact_amd import capi, scenes
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


def import_through_libdrm(exp, tmp, tag):
    """the dma-buf leg: a process without HIP (tests/cpp/dmabuf_import: libdrm_amdgpu's amdgpu_bo_import + amdgpu_bo_cpu_map, the kernel graphics
    driver's own API — what a Vulkan driver does underneath VK_EXT_external_memory_dma_buf) inherits the descriptor, imports it and returns the
    buffer's bytes [dmabuf_offset, dmabuf_offset + bytes)"""
    exe = os.path.join(HERE, "cpp", "dmabuf_import")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp"), "dmabuf_import"])
    got = {}
    for n in NAMES:
        fd = exp[n]["dmabuf_fd"]
        assert fd >= 0, f"ivx_mesh_export made no dma-buf descriptor for {n} on this box: {capi.lib().ivx_last_error().decode()}"
        assert "dmabuf" in os.readlink(f"/proc/self/fd/{fd}")
        assert exp[n]["dmabuf_bytes"] >= exp[n]["dmabuf_offset"] % 4096 + exp[n]["capacity_bytes"]
        out = os.path.join(tmp, f"drm_{tag}_{n}.bin")
        r = subprocess.run([exe, str(fd), str(exp[n]["dmabuf_offset"]), str(exp[n]["bytes"]), out], pass_fds=(fd,), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, f"non-HIP import of {n} failed:\n{r.stderr[-3000:]}"
        got[n] = np.fromfile(out, dtype=np.uint8)
        os.close(fd)
    return got


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
            # ... and the same bytes through the dma-buf descriptor, imported WITHOUT HIP (the renderer's leg)
            drm = import_through_libdrm(exp, tmp, "full")
            for n in NAMES:
                np.testing.assert_array_equal(drm[n], np.ascontiguousarray(want[n]).view(np.uint8).reshape(-1), err_msg=n + " through the dma-buf descriptor")
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
            drm2 = import_through_libdrm(exp2, tmp, "sync")
            for n, w in zip(NAMES, (pos, nrm, idx, im, sub)):
                wb = np.ascontiguousarray(w).view(np.uint8).reshape(-1)
                np.testing.assert_array_equal(drm2[n][:wb.size], wb, err_msg=n + " after sync, through the dma-buf descriptor")
    finally:
        g.close()
