#!/usr/bin/env python3
"""The importing side of tests/test_gpu_mesh_export.py: a fresh process (a stand-in for a renderer) that opens the mesh buffers another
process exported (hipIpcOpenMemHandle on the 64-byte handles of ivx_mesh_export), copies `bytes` of each to the host and writes them to <out>.
usage: mesh_import_worker.py <request.npz> <out.npz>"""
import ctypes as C
import sys

import numpy as np


import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from impact_amd import capi

    req = np.load(sys.argv[1])
    lib = capi.lib()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = {}
    for name in ("positions", "normals", "indices", "index_materials", "submeshes"):
        handle = np.ascontiguousarray(req[name + "_handle"], dtype=np.uint8)
        n = int(req[name + "_bytes"])
        dev = C.c_void_p()
        rc = lib.ivx_mesh_import_open(handle.ctypes.data_as(C.c_void_p), 0, C.byref(dev))
        assert rc == 0, f"ivx_mesh_import_open({name}) -> {rc}: {lib.ivx_last_error().decode()}"
        buf = np.zeros(n, dtype=np.uint8)
        assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), dev, n, 2) == 0
        assert lib.ivx_mesh_import_close(dev) == 0
        out[name] = buf
    np.savez(sys.argv[2], **out)


if __name__ == "__main__":
    main()
