"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker. The product package `impact_amd` never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

CHUNK_INFO_DTYPE = np.dtype(
    [
        ("kind", "u1"),
        ("gen_kind", "u1"),
        ("flags", "u1"),
        ("uniform_type", "u1"),
        ("face_dist", "<u2"),
        ("region_count", "u1"),
        ("boundary_region_count", "u1"),
    ]
)
assert CHUNK_INFO_DTYPE.itemsize == 8


def build_oracle(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, "src", f) for f in os.listdir(os.path.join(ORACLE_DIR, "src"))]
    srcs.append(os.path.join(ORACLE_DIR, "include", "oracle.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.orc_sdf_compile.restype = C.c_int
        L.orc_sdf_compile.argtypes = [vp, C.c_int, C.c_uint32, vp, C.c_int, vp, vp]
        L.orc_object_from_sdf.restype = vp
        L.orc_object_from_sdf.argtypes = [vp, C.c_int, C.c_uint32, C.c_float, C.c_uint8]
        L.orc_object_from_sdf_parallel.restype = vp
        L.orc_object_from_sdf_parallel.argtypes = [vp, C.c_int, C.c_uint32, C.c_float, C.c_uint8, C.c_int]
        L.orc_mesh_recreate_parallel.restype = vp
        L.orc_mesh_recreate_parallel.argtypes = [vp, C.c_int]
        L.orc_inertia_parallel.argtypes = [vp, vp, vp, C.c_int]
        L.orc_object_from_box.restype = vp
        L.orc_object_from_box.argtypes = [vp, vp, C.c_uint8, C.c_int8, C.c_uint8]
        L.orc_object_from_manual.restype = vp
        L.orc_object_from_manual.argtypes = [C.c_int, vp, vp]
        L.orc_object_from_dense.restype = vp
        L.orc_object_from_dense.argtypes = [vp, C.c_float, vp, vp]
        L.orc_object_free.argtypes = [vp]
        L.orc_update_occupied_voxel_ranges.argtypes = [vp]
        L.orc_compute_all_derived_state.argtypes = [vp]
        L.orc_object_info.argtypes = [vp, vp]
        L.orc_object_extent.restype = C.c_float
        L.orc_object_extent.argtypes = [vp]
        L.orc_export_dense.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_export_sparse.argtypes = [vp, vp, vp, vp]
        L.orc_mesh_recreate.restype = vp
        L.orc_mesh_recreate.argtypes = [vp]
        L.orc_mesh_counts.argtypes = [vp, vp]
        L.orc_mesh_get.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_mesh_free.argtypes = [vp]
        L.orc_chunk_sdf.restype = C.c_int
        L.orc_chunk_sdf.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
        L.orc_vertex_materials.argtypes = [vp, vp, vp, vp]
        L.orc_index_materials.argtypes = [vp, vp, vp]
        L.orc_inertia.argtypes = [vp, vp, vp, vp]
        L.orc_derive_inertial_properties.argtypes = [vp, vp]
        L.orc_region_labels.restype = C.c_uint32
        L.orc_region_labels.argtypes = [vp, vp]
        L.orc_sd_from_f32.restype = C.c_int8
        L.orc_sd_from_f32.argtypes = [C.c_float]
        L.orc_sd_to_f32.restype = C.c_float
        L.orc_sd_to_f32.argtypes = [C.c_int8]
        L.orc_split_off_smallest_region.restype = C.c_int
        L.orc_split_off_smallest_region.argtypes = [vp, C.POINTER(vp), vp]
        L.orc_clip_polyhedron.restype = C.c_int
        L.orc_clip_polyhedron.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.POINTER(vp), vp]
        L.orc_sphere_voxel_object_contacts.restype = C.c_int
        L.orc_sphere_voxel_object_contacts.argtypes = [vp, vp, vp, vp, C.c_float, C.c_int, vp, vp, vp, vp]
        L.orc_collision_probes.restype = C.c_int
        L.orc_collision_probes.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.POINTER(C.c_uint32)]
        L.orc_mutual_voxel_object_contacts.restype = C.c_int
        L.orc_mutual_voxel_object_contacts.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, vp, C.c_int, vp, vp, vp, vp]
        L.orc_capsule_voxel_object_contacts.restype = C.c_int
        L.orc_capsule_voxel_object_contacts.argtypes = [vp, vp, vp, vp, vp, C.c_float, C.c_int, vp, vp, vp, vp]
        L.orc_plane_voxel_object_contacts.restype = C.c_int
        L.orc_plane_voxel_object_contacts.argtypes = [vp, vp, vp, vp, C.c_float, C.c_int, vp, vp, vp, vp]
        L.orc_offset_reference_point.restype = None
        L.orc_offset_reference_point.argtypes = [vp, vp]
        L.orc_apply_updated_inertial_properties.restype = None
        L.orc_apply_updated_inertial_properties.argtypes = [vp, vp, vp, C.c_int, vp]
        L.orc_extracted_object_dynamics.restype = None
        L.orc_extracted_object_dynamics.argtypes = [vp, vp, C.c_float, vp, vp, vp, vp]
        L.orc_mesh_modifications.restype = C.c_int
        L.orc_mesh_modifications.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
        L.orc_mesh_report_synchronized.restype = None
        L.orc_mesh_report_synchronized.argtypes = [vp]
        L.orc_probes_recompute.restype = C.c_void_p
        L.orc_probes_recompute.argtypes = [vp, vp]
        L.orc_probes_sync.restype = None
        L.orc_probes_sync.argtypes = [vp, vp, vp, vp]
        L.orc_probes_get.restype = C.c_uint32
        L.orc_probes_get.argtypes = [vp, vp, C.c_uint32, vp, C.POINTER(C.c_uint32)]
        L.orc_probes_free.restype = None
        L.orc_probes_free.argtypes = [vp]
        L.orc_mesh_sync.restype = None
        L.orc_mesh_sync.argtypes = [vp, vp, vp]
        L.orc_range_allocator_script.restype = None
        L.orc_range_allocator_script.argtypes = [vp, C.c_int, vp]
        L.orc_absorb_mutual.restype = None
        L.orc_absorb_mutual.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp]
        L.orc_absorb_capsule.restype = C.c_int
        L.orc_absorb_capsule.argtypes = [vp, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, C.POINTER(C.c_uint32)]
        L.orc_absorb_sphere.restype = C.c_int
        L.orc_absorb_sphere.argtypes = [vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, C.POINTER(C.c_uint32)]
        L.orc_physics_create.restype = vp
        L.orc_physics_free.argtypes = [vp]
        L.orc_physics_set_config.argtypes = [vp, vp]
        L.orc_physics_set_bodies.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.orc_physics_get_bodies.argtypes = [vp, vp, vp]
        L.orc_rigid_body_new.argtypes = [vp, C.c_float, vp, vp, vp, vp, vp, vp]
        L.orc_rigid_body_motion.argtypes = [vp, vp, vp]
        L.orc_sphere_sphere_contact.restype = C.c_int
        L.orc_sphere_sphere_contact.argtypes = [vp, C.c_float, vp, C.c_float, vp, vp, vp]
        L.orc_sphere_plane_contact.restype = C.c_int
        L.orc_sphere_plane_contact.argtypes = [vp, C.c_float, vp, C.c_float, vp, vp, vp]
        L.orc_physics_prepare.restype = C.c_int
        L.orc_physics_prepare.argtypes = [vp, vp, C.c_int]
        L.orc_physics_set_joints.argtypes = [vp, vp, C.c_int]
        L.orc_physics_prepared_body_count.restype = C.c_int
        L.orc_physics_prepared_body_count.argtypes = [vp]
        L.orc_physics_contact_order.argtypes = [vp, vp]
        L.orc_physics_accumulated_impulses.argtypes = [vp, vp]
        L.orc_physics_advance_momenta.argtypes = [vp, C.c_float]
        L.orc_physics_solve.argtypes = [vp]
        L.orc_physics_advance_configurations.argtypes = [vp, C.c_float]
        L.orc_physics_step.restype = C.c_int
        L.orc_physics_step.argtypes = [vp, vp, C.c_int, C.c_float]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def tiled_to_dense(a: np.ndarray, cc) -> np.ndarray:
    """chunk-tiled (n_chunks*4096) -> x-major dense (nx,ny,nz)."""
    cx, cy, cz = cc
    return a.reshape(cx, cy, cz, 16, 16, 16).transpose(0, 3, 1, 4, 2, 5).reshape(cx * 16, cy * 16, cz * 16)


def dense_to_tiled(a: np.ndarray) -> np.ndarray:
    nx, ny, nz = a.shape
    cx, cy, cz = nx // 16, ny // 16, nz // 16
    return np.ascontiguousarray(a.reshape(cx, 16, cy, 16, cz, 16).transpose(0, 2, 4, 1, 3, 5)).reshape(-1)


def canonicalize_labels(lab: np.ndarray, empty) -> np.ndarray:
    """Relabel components by order of first occurrence in the flattened array (empty -> 0xFFFFFFFF)."""
    flat = lab.reshape(-1)
    out = np.full(flat.shape, 0xFFFFFFFF, dtype=np.uint32)
    mask = flat != empty
    vals = flat[mask]
    uniq, first = np.unique(vals, return_index=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    out[mask] = rank[np.searchsorted(uniq, vals)].astype(np.uint32)
    return out.reshape(lab.shape)


class OracleMesh:
    def __init__(self, positions, normals, indices, index_materials, submeshes):
        self.positions, self.normals, self.indices = positions, normals, indices
        self.index_materials, self.submeshes = index_materials, submeshes


class OracleMeshHandle:
    """a VoxelObjectMesh that stays alive: recreate once, then sync_with_voxel_object after edits (mesh.rs:286-456)"""

    def __init__(self, obj: "OracleObject"):
        self.obj = obj
        self.h = C.c_void_p(lib().orc_mesh_recreate(obj.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mesh_free(self.h)
            self.h = None

    def sync(self, invalidated):
        inv = np.ascontiguousarray(invalidated, dtype=np.uint8)
        lib().orc_mesh_sync(self.h, self.obj.h, _p(inv))

    def modifications(self):
        removed = C.c_int(0)
        n = lib().orc_mesh_modifications(self.h, None, 0, C.byref(removed))
        out = np.zeros((max(1, n), 4), dtype=np.uint32)
        lib().orc_mesh_modifications(self.h, _p(out), n, C.byref(removed))
        return out[:n], bool(removed.value)

    def report_synchronized(self):
        lib().orc_mesh_report_synchronized(self.h)

    def get(self) -> "OracleMesh":
        L = lib()
        cnt = np.zeros(3, dtype=np.uint32)
        L.orc_mesh_counts(self.h, _p(cnt))
        nv, ni, ns = (int(x) for x in cnt)
        pos = np.empty((nv, 3), dtype=np.float32)
        nrm = np.empty((nv, 3), dtype=np.float32)
        idx = np.empty(ni, dtype=np.uint32)
        im = np.empty((ni, 8), dtype=np.uint8)
        sub = np.empty((ns, 16), dtype=np.uint32)
        L.orc_mesh_get(self.h, _p(pos), _p(nrm), _p(idx), _p(im), _p(sub))
        return OracleMesh(pos, nrm, idx, im, sub)


class OracleProbes:
    """VoxelObjectCollisionProbes kept alive next to an OracleMeshHandle: recompute, then sync after every mesh sync"""

    def __init__(self, mesh: OracleMeshHandle):
        self.mesh = mesh
        self.h = C.c_void_p(lib().orc_probes_recompute(mesh.obj.h, mesh.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_probes_free(self.h)
            self.h = None

    def sync(self, invalidated):
        lib().orc_probes_sync(self.h, self.mesh.obj.h, self.mesh.h, _p(np.ascontiguousarray(invalidated, dtype=np.uint8)))

    def get(self):
        """-> (points [n,3] incl. freed ranges, entries [m,5] sorted by range start)"""
        n = lib().orc_probes_get(self.h, None, 0, None, None)
        cc = self.mesh.obj.chunk_counts
        pts = np.zeros((max(1, n), 3), dtype=np.float32)
        ent = np.zeros((cc[0] * cc[1] * cc[2], 5), dtype=np.uint32)
        ne = C.c_uint32(0)
        lib().orc_probes_get(self.h, _p(pts), n, _p(ent), C.byref(ne))
        return pts[:n], ent[: ne.value].copy()


def range_allocator_script(ops):
    """ops: list of ("free", a, b) | ("alloc", n) | ("merge",) -> list of results (alloc: (start, end) or None)"""
    code = {"free": 0, "alloc": 1, "merge": 2}
    arr = np.zeros((len(ops), 3), dtype=np.int64)
    for i, op in enumerate(ops):
        arr[i, 0] = code[op[0]]
        for k, v in enumerate(op[1:]):
            arr[i, 1 + k] = v
    res = np.zeros((len(ops), 2), dtype=np.int64)
    lib().orc_range_allocator_script(_p(arr), len(ops), _p(res))
    return [((int(r[0]), int(r[1])) if r[0] >= 0 else None) if op[0] == "alloc" else None for op, r in zip(ops, res)]


class OracleObject:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle object construction failed")
        self.h = C.c_void_p(handle)

    def __del__(self):
        try:
            if self.h:
                lib().orc_object_free(self.h)
                self.h = None
        except Exception:
            pass

    @classmethod
    def from_sdf(cls, graph, voxel_extent=1.0, voxel_type=0):
        nodes = graph.nodes()
        return cls(lib().orc_object_from_sdf(_p(nodes), len(nodes), graph.root_node_id, voxel_extent, voxel_type))

    @classmethod
    def from_sdf_parallel(cls, graph, voxel_extent=1.0, voxel_type=0, threads=2):
        """generate + occupied ranges + derived state over `threads` OpenMP threads (same object as the sequential path)"""
        nodes = graph.nodes()
        return cls(lib().orc_object_from_sdf_parallel(_p(nodes), len(nodes), graph.root_node_id, voxel_extent, voxel_type, threads))

    @classmethod
    def from_box(cls, shape, offset=(0, 0, 0), voxel=(0, -128, 0)):
        s = np.asarray(shape, dtype=np.int32)
        o = np.asarray(offset, dtype=np.int32)
        return cls(lib().orc_object_from_box(_p(s), _p(o), voxel[0], voxel[1], voxel[2]))

    @classmethod
    def from_manual(cls, cells, offset=(0, 0, 0)):
        c = np.ascontiguousarray(np.asarray(cells, dtype=np.uint8))
        n = c.shape[0]
        assert c.shape == (n, n, n)
        o = np.asarray(offset, dtype=np.int32)
        return cls(lib().orc_object_from_manual(n, _p(c), _p(o)))

    @classmethod
    def from_dense(cls, cc, sdf_tiled, type_tiled, voxel_extent=1.0):
        ccs = np.asarray(cc, dtype=np.int32)
        sdf_tiled = np.ascontiguousarray(sdf_tiled, dtype=np.int8)
        type_tiled = np.ascontiguousarray(type_tiled, dtype=np.uint8)
        return cls(lib().orc_object_from_dense(_p(ccs), voxel_extent, _p(sdf_tiled), _p(type_tiled)))

    def update_occupied_voxel_ranges(self):
        lib().orc_update_occupied_voxel_ranges(self.h)

    def compute_all_derived_state(self):
        lib().orc_compute_all_derived_state(self.h)

    def info(self):
        out = np.zeros(19, dtype=np.int32)
        lib().orc_object_info(self.h, _p(out))
        return {
            "chunk_counts": tuple(int(x) for x in out[0:3]),
            "stored_chunks": int(out[3]),
            "occupied_chunk_ranges": [(int(out[4 + 2 * d]), int(out[5 + 2 * d])) for d in range(3)],
            "occupied_voxel_ranges": [(int(out[10 + 2 * d]), int(out[11 + 2 * d])) for d in range(3)],
            "grid_shape": tuple(int(x) for x in out[16:19]),
        }

    @property
    def chunk_counts(self):
        return self.info()["chunk_counts"]

    def export_dense(self):
        cc = self.chunk_counts
        n = cc[0] * cc[1] * cc[2]
        sdf = np.empty(n * 4096, dtype=np.int8)
        typ = np.empty(n * 4096, dtype=np.uint8)
        flg = np.empty(n * 4096, dtype=np.uint8)
        lab = np.empty(n * 4096, dtype=np.uint8)
        info = np.zeros(n, dtype=CHUNK_INFO_DTYPE)
        lib().orc_export_dense(self.h, _p(sdf), _p(typ), _p(flg), _p(lab), _p(info))
        return sdf, typ, flg, lab, info

    def export_sparse(self):
        inf = self.info()
        cc = inf["chunk_counts"]
        n = cc[0] * cc[1] * cc[2]
        offs = np.empty(n, dtype=np.int32)
        vox = np.empty((inf["stored_chunks"] * 4096, 3), dtype=np.uint8)
        lab = np.empty(inf["stored_chunks"] * 4096, dtype=np.uint8)
        lib().orc_export_sparse(self.h, _p(offs), _p(vox), _p(lab))
        return offs, vox, lab

    def voxel_flags(self, i, j, k):
        """flags of the voxel at object indices, or None if empty (get_voxel_if_occupied)."""
        cc = self.chunk_counts
        _, _, flg, _, _ = self.export_dense()
        c = ((i >> 4) * cc[1] + (j >> 4)) * cc[2] + (k >> 4)
        f = int(flg[c * 4096 + (((i & 15) << 8) | ((j & 15) << 4) | (k & 15))])
        return None if (f & 1) else f

    def mesh_parallel(self, threads) -> OracleMesh:
        return self.mesh(threads)

    def inertia_parallel(self, threads, densities=None):
        d = np.ones(256, dtype=np.float32) if densities is None else np.ascontiguousarray(densities, dtype=np.float32)
        o32 = np.zeros(10, dtype=np.float32)
        lib().orc_inertia_parallel(self.h, _p(d), _p(o32), threads)
        return o32

    def mesh(self, threads=1) -> OracleMesh:
        import time

        L = lib()
        t0 = time.perf_counter()
        m = C.c_void_p(L.orc_mesh_recreate_parallel(self.h, threads) if threads > 1 else L.orc_mesh_recreate(self.h))
        self.last_mesh_seconds = time.perf_counter() - t0  # the meshing alone, without the export into numpy arrays below
        cnt = np.zeros(3, dtype=np.uint32)
        L.orc_mesh_counts(m, _p(cnt))
        nv, ni, ns = (int(x) for x in cnt)
        pos = np.empty((nv, 3), dtype=np.float32)
        nrm = np.empty((nv, 3), dtype=np.float32)
        idx = np.empty(ni, dtype=np.uint32)
        im = np.empty((ni, 8), dtype=np.uint8)
        sub = np.empty((ns, 16), dtype=np.uint32)
        L.orc_mesh_get(m, _p(pos), _p(nrm), _p(idx), _p(im), _p(sub))
        L.orc_mesh_free(m)
        return OracleMesh(pos, nrm, idx, im, sub)

    def chunk_sdf(self, ci, cj, ck):
        """padded 18^3 (values f32, types u8) of an exposed non-uniform chunk, or None"""
        val = np.empty((18, 18, 18), dtype=np.float32)
        typ = np.empty((18, 18, 18), dtype=np.uint8)
        if not lib().orc_chunk_sdf(self.h, ci, cj, ck, _p(val), _p(typ)):
            return None
        return val, typ

    def split_off_smallest_region(self):
        """-> (outcome, child OracleObject or None, origin offset in parent voxels)"""
        child = C.c_void_p()
        origin = np.zeros(3, dtype=np.int32)
        rc = lib().orc_split_off_smallest_region(self.h, C.byref(child), _p(origin))
        return rc, (OracleObject(child.value) if rc == 1 else None), tuple(int(x) for x in origin)

    def clip_polyhedron(self, planes, aabb, copy=False):
        """extract_polyhedron / copy_polyhedron -> (outcome, child or None, origin offset)"""
        pl = np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, 4)
        bb = np.ascontiguousarray(aabb, dtype=np.float32).reshape(6)
        child = C.c_void_p()
        origin = np.zeros(3, dtype=np.int32)
        rc = lib().orc_clip_polyhedron(self.h, _p(pl), len(pl), _p(bb), 1 if copy else 0, C.byref(child), _p(origin))
        return rc, (OracleObject(child.value) if rc == 1 else None), tuple(int(x) for x in origin)

    def absorb_sphere(self, center, influence_radius, sphere_radius, densities=None):
        """apply_sphere_absorption with the sphere in the object's normalized space -> dict(removed64, emptied_by_type,
        invalidated (bool per chunk), touched_chunks, removed_chunks)"""
        d = np.ones(256, dtype=np.float32) if densities is None else np.ascontiguousarray(densities, dtype=np.float32)
        c = np.ascontiguousarray(center, dtype=np.float32)
        cc = self.chunk_counts
        removed = np.zeros(10, dtype=np.float64)
        by_type = np.zeros(256, dtype=np.uint32)
        inval = np.zeros(cc[0] * cc[1] * cc[2], dtype=np.uint8)
        touched = C.c_uint32(0)
        n = lib().orc_absorb_sphere(self.h, _p(c), influence_radius, sphere_radius, _p(d), _p(removed), _p(by_type), _p(inval), C.byref(touched))
        return {"removed64": removed, "emptied_by_type": by_type, "invalidated": inval.astype(bool), "touched_chunks": int(touched.value),
                "removed_chunks": int(n)}

    def absorb_capsule(self, segment_start, segment_vector, influence_radius, capsule_radius, densities=None):
        """apply_capsule_absorption with the capsule in the object's normalized space; same result dict as absorb_sphere"""
        d = np.ones(256, dtype=np.float32) if densities is None else np.ascontiguousarray(densities, dtype=np.float32)
        a = np.ascontiguousarray(segment_start, dtype=np.float32)
        v = np.ascontiguousarray(segment_vector, dtype=np.float32)
        cc = self.chunk_counts
        removed = np.zeros(10, dtype=np.float64)
        by_type = np.zeros(256, dtype=np.uint32)
        inval = np.zeros(cc[0] * cc[1] * cc[2], dtype=np.uint8)
        touched = C.c_uint32(0)
        n = lib().orc_absorb_capsule(self.h, _p(a), _p(v), influence_radius, capsule_radius, _p(d), _p(removed), _p(by_type), _p(inval), C.byref(touched))
        return {"removed64": removed, "emptied_by_type": by_type, "invalidated": inval.astype(bool), "touched_chunks": int(touched.value),
                "removed_chunks": int(n)}

    def absorb_mutual(self, rotation_xyzw, translation, other, other_rotation_xyzw, other_translation, smoothness, densities=None, other_densities=None):
        """apply_mutual_absorption(self = A, other = B), world -> object transforms -> (result dict of A, result dict of B)"""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
        da = np.ones(256, dtype=np.float32) if densities is None else f(densities)
        db = np.ones(256, dtype=np.float32) if other_densities is None else f(other_densities)
        ra, rb = np.zeros(10, dtype=np.float64), np.zeros(10, dtype=np.float64)
        cca, ccb = self.chunk_counts, other.chunk_counts
        ia, ib = np.zeros(cca[0] * cca[1] * cca[2], dtype=np.uint8), np.zeros(ccb[0] * ccb[1] * ccb[2], dtype=np.uint8)
        st = np.zeros(6, dtype=np.uint64)
        lib().orc_absorb_mutual(self.h, _p(f(rotation_xyzw)), _p(f(translation)), _p(da), other.h, _p(f(other_rotation_xyzw)), _p(f(other_translation)),
                                _p(db), smoothness, _p(ra), _p(rb), _p(ia), _p(ib), _p(st))
        mk = lambda r, inv, o: {"removed64": r, "invalidated": inv.astype(bool), "emptied_voxels": int(st[o]), "touched_chunks": int(st[o + 1]),  # noqa: E731
                                "removed_chunks": int(st[o + 2])}
        return mk(ra, ia, 0), mk(rb, ib, 3)

    def sphere_contacts(self, rotation_xyzw, translation, center, radius, cap=65536):
        """for_each_sphere_voxel_object_contact -> (indices [n,3], position [n,3], normal [n,3], depth [n])"""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
        idx = np.zeros((cap, 3), dtype=np.int32)
        pos = np.zeros((cap, 3), dtype=np.float32)
        nrm = np.zeros((cap, 3), dtype=np.float32)
        dep = np.zeros(cap, dtype=np.float32)
        n = lib().orc_sphere_voxel_object_contacts(self.h, _p(f(rotation_xyzw)), _p(f(translation)), _p(f(center)), radius, cap, _p(idx), _p(pos), _p(nrm),
                                                   _p(dep))
        assert n <= cap
        return idx[:n], pos[:n], nrm[:n], dep[:n]

    def collision_probes(self, mesh):
        """VoxelObjectCollisionProbes::recompute_for_all_chunks on an OracleMesh -> (points [n,3] f32, entries [m,5] u32: chunk i,j,k,
        first point, end point)"""
        ns = len(mesh.submeshes)
        cap = max(1, ns * 4096)
        while True:
            pts = np.zeros((cap, 3), dtype=np.float32)
            ent = np.zeros((max(1, ns), 5), dtype=np.uint32)
            ne = C.c_uint32(0)
            n = lib().orc_collision_probes(self.h, _p(np.ascontiguousarray(mesh.positions)), _p(np.ascontiguousarray(mesh.normals)),
                                           _p(np.ascontiguousarray(mesh.indices)), _p(np.ascontiguousarray(mesh.submeshes)), ns, _p(pts), cap, _p(ent),
                                           C.byref(ne))
            if n <= cap:
                return pts[:n].copy(), ent[: ne.value].copy()
            cap = n

    def center_of_mass(self, densities=None):
        """derive_center_of_mass (object/inertia.rs:167-169): first moments / mass, Vector3 / f32 = multiply by the reciprocal"""
        m32 = self.inertia(densities)[0]
        return (m32[1:4] * (np.float32(1.0) / m32[0])).astype(np.float32)

    def mutual_contacts(self, probes, com, rotation_xyzw, translation, other, other_probes, other_com, other_rotation_xyzw, other_translation, cap=65536):
        """for_each_mutual_voxel_object_contact(self = A, other = B) -> (which_ijk [n,4] i32, position, normal [n,3], depth [n])"""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
        pa, ea = probes
        pb, eb = other_probes
        wi = np.zeros((cap, 4), dtype=np.int32)
        pos = np.zeros((cap, 3), dtype=np.float32)
        nrm = np.zeros((cap, 3), dtype=np.float32)
        dep = np.zeros(cap, dtype=np.float32)
        n = lib().orc_mutual_voxel_object_contacts(self.h, _p(f(pa)), _p(np.ascontiguousarray(ea, dtype=np.uint32)), len(ea), _p(f(com)), _p(f(rotation_xyzw)),
                                                   _p(f(translation)), other.h, _p(f(pb)), _p(np.ascontiguousarray(eb, dtype=np.uint32)), len(eb),
                                                   _p(f(other_com)), _p(f(other_rotation_xyzw)), _p(f(other_translation)), cap, _p(wi), _p(pos), _p(nrm),
                                                   _p(dep))
        assert n <= cap
        return wi[:n], pos[:n], nrm[:n], dep[:n]

    def capsule_contacts(self, rotation_xyzw, translation, segment_start, segment_vector, radius, cap=65536):
        """for_each_capsule_voxel_object_contact -> (indices [n,3], position [n,3], normal [n,3], depth [n])"""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
        idx = np.zeros((cap, 3), dtype=np.int32)
        pos = np.zeros((cap, 3), dtype=np.float32)
        nrm = np.zeros((cap, 3), dtype=np.float32)
        dep = np.zeros(cap, dtype=np.float32)
        n = lib().orc_capsule_voxel_object_contacts(self.h, _p(f(rotation_xyzw)), _p(f(translation)), _p(f(segment_start)), _p(f(segment_vector)), radius, cap,
                                                    _p(idx), _p(pos), _p(nrm), _p(dep))
        assert n <= cap
        return idx[:n], pos[:n], nrm[:n], dep[:n]

    def plane_contacts(self, rotation_xyzw, translation, normal, displacement, cap=65536):
        """for_each_voxel_object_plane_contact -> (indices [n,3], position [n,3], normal [n,3], depth [n])"""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
        idx = np.zeros((cap, 3), dtype=np.int32)
        pos = np.zeros((cap, 3), dtype=np.float32)
        nrm = np.zeros((cap, 3), dtype=np.float32)
        dep = np.zeros(cap, dtype=np.float32)
        n = lib().orc_plane_voxel_object_contacts(self.h, _p(f(rotation_xyzw)), _p(f(translation)), _p(f(normal)), displacement, cap, _p(idx), _p(pos),
                                                  _p(nrm), _p(dep))
        assert n <= cap
        return idx[:n], pos[:n], nrm[:n], dep[:n]

    def inertia(self, densities=None):
        d = np.ones(256, dtype=np.float32) if densities is None else np.ascontiguousarray(densities, dtype=np.float32)
        o32 = np.zeros(10, dtype=np.float32)
        o64 = np.zeros(10, dtype=np.float64)
        lib().orc_inertia(self.h, _p(d), _p(o32), _p(o64))
        return o32, o64

    def region_labels(self, want_labels=True):
        cc = self.chunk_counts
        lab = np.empty((cc[0] * 16, cc[1] * 16, cc[2] * 16), dtype=np.uint32) if want_labels else None
        n = lib().orc_region_labels(self.h, _p(lab))
        return int(n), lab


# ---- a14: rigid bodies after voxel removal (impact_voxel/src/interaction.rs:405-602) ------------------------------------------------
def offset_reference_point(moments32, offset):
    m = np.ascontiguousarray(moments32, dtype=np.float32).copy()
    lib().orc_offset_reference_point(_p(m), _p(np.ascontiguousarray(offset, dtype=np.float32)))
    return m


def apply_updated_inertial_properties(body, moments32, original_local_com, preserve_momentum=False):
    from impact_amd.capi import RIGID_BODY_DTYPE

    b = np.ascontiguousarray(body, dtype=RIGID_BODY_DTYPE).reshape(1).copy()
    com = np.zeros(3, dtype=np.float32)
    lib().orc_apply_updated_inertial_properties(_p(b), _p(np.ascontiguousarray(moments32, dtype=np.float32)),
                                                _p(np.ascontiguousarray(original_local_com, dtype=np.float32)), 1 if preserve_momentum else 0, _p(com))
    return b[0], com


def extracted_object_dynamics(moments32_in_parent_frame, origin_offset_in_parent, voxel_extent, original_local_com, parent_body):
    from impact_amd.capi import RIGID_BODY_DTYPE

    m = np.ascontiguousarray(moments32_in_parent_frame, dtype=np.float32).copy()
    pb = np.ascontiguousarray(parent_body, dtype=RIGID_BODY_DTYPE).reshape(1)
    fb = np.zeros(1, dtype=RIGID_BODY_DTYPE)
    com = np.zeros(3, dtype=np.float32)
    lib().orc_extracted_object_dynamics(_p(m), _p(np.ascontiguousarray(origin_offset_in_parent, dtype=np.int32)), voxel_extent,
                                        _p(np.ascontiguousarray(original_local_com, dtype=np.float32)), _p(pb), _p(fb), _p(com))
    return fb[0], m, com


def sdf_compile(graph):
    from impact_amd.sdf_graph import PROCESSED_NODE_DTYPE

    nodes = graph.nodes()
    out = np.zeros(4 * max(1, len(nodes)) + 64, dtype=PROCESSED_NODE_DTYPE)
    dom = np.zeros(6, dtype=np.float32)
    ss = C.c_int(0)
    n = lib().orc_sdf_compile(_p(nodes), len(nodes), graph.root_node_id, _p(out), len(out), _p(dom), C.byref(ss))
    if n < 0:
        raise RuntimeError("oracle sdf compile failed")
    return out[:n].copy(), dom, ss.value


def derive_inertial_properties(moments32):
    m = np.ascontiguousarray(moments32, dtype=np.float32)
    out = np.zeros(22, dtype=np.float32)
    lib().orc_derive_inertial_properties(_p(m), _p(out))
    return {"mass": float(out[0]), "com": out[1:4].copy(), "inertia": out[4:13].reshape(3, 3).T.copy(), "inverse": out[13:22].reshape(3, 3).T.copy()}


# ---- rigid bodies + contact solver --------------------------------------------------------------
def rigid_body_new(mass, inertia, position, orientation=(0, 0, 0, 1), velocity=(0, 0, 0), angular_velocity=(0, 0, 0)):
    """DynamicRigidBody::new (rigid_body.rs:411-441); inertia = 3x3 about the centre of mass (body frame)"""
    from impact_amd.capi import RIGID_BODY_DTYPE

    out = np.zeros(1, dtype=RIGID_BODY_DTYPE)
    I = np.asarray(inertia, dtype=np.float64).reshape(3, 3)
    Ic = np.ascontiguousarray(I.T.reshape(-1), dtype=np.float32)  # column-major
    Iinv = np.ascontiguousarray(np.linalg.inv(I).T.reshape(-1), dtype=np.float32)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    lib().orc_rigid_body_new(_p(out), float(mass), _p(Ic), _p(Iinv), _p(f(position)), _p(f(orientation)), _p(f(velocity)), _p(f(angular_velocity)))
    return out[0]


def uniform_sphere_body(radius, density, position, velocity=(0, 0, 0)):
    """InertialProperties::of_uniform_sphere (inertia.rs:155-168) in f32 like the reference"""
    f32 = np.float32
    mass = f32(f32(4.0 / 3.0) * f32(np.pi) * f32(radius) ** 3) * f32(density)
    moi = f32(f32(2.0 / 5.0) * mass) * f32(radius) ** 2
    return rigid_body_new(float(mass), np.eye(3) * float(moi), position, velocity=velocity)


def body_motion(body):
    v = np.zeros(3, dtype=np.float32)
    w = np.zeros(3, dtype=np.float32)
    b = np.array([body])
    lib().orc_rigid_body_motion(_p(b), _p(v), _p(w))
    return v, w


def sphere_sphere_contact(ca, ra, cb, rb):
    pos, nrm, d = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_float(0)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    if not lib().orc_sphere_sphere_contact(_p(f(ca)), ra, _p(f(cb)), rb, _p(pos), _p(nrm), C.byref(d)):
        return None
    return pos, nrm, d.value


def sphere_plane_contact(c, r, plane_normal=(0, 1, 0), displacement=0.0):
    pos, nrm, d = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_float(0)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    if not lib().orc_sphere_plane_contact(_p(f(c)), r, _p(f(plane_normal)), displacement, _p(pos), _p(nrm), C.byref(d)):
        return None
    return pos, nrm, d.value


class OraclePhysics:
    """RigidBodyManager + ConstraintManager of the reference, restated (oracle/src/orc_physics.cpp)"""

    def __init__(self, dynamic, kinematic=None, config=None):
        from impact_amd.capi import KINEMATIC_BODY_DTYPE, RIGID_BODY_DTYPE, SOLVER_CONFIG_DTYPE

        self.h = C.c_void_p(lib().orc_physics_create())
        self.n_dyn = len(dynamic)
        kin = np.zeros(0, dtype=KINEMATIC_BODY_DTYPE) if kinematic is None else np.ascontiguousarray(kinematic, dtype=KINEMATIC_BODY_DTYPE)
        self.n_kin = len(kin)
        dyn = np.ascontiguousarray(dynamic, dtype=RIGID_BODY_DTYPE)
        lib().orc_physics_set_bodies(self.h, _p(dyn), len(dyn), _p(kin), len(kin))
        if config is not None:
            cfg = np.zeros(1, dtype=SOLVER_CONFIG_DTYPE)
            cfg[0] = config
            lib().orc_physics_set_config(self.h, _p(cfg))

    def __del__(self):
        try:
            lib().orc_physics_free(self.h)
        except Exception:
            pass

    def bodies(self):
        from impact_amd.capi import KINEMATIC_BODY_DTYPE, RIGID_BODY_DTYPE

        dyn = np.zeros(self.n_dyn, dtype=RIGID_BODY_DTYPE)
        kin = np.zeros(self.n_kin, dtype=KINEMATIC_BODY_DTYPE)
        lib().orc_physics_get_bodies(self.h, _p(dyn), _p(kin))
        return dyn, kin

    def prepare(self, contacts):
        c = np.ascontiguousarray(contacts)
        self.n_prepared = lib().orc_physics_prepare(self.h, _p(c), len(c))
        return self.n_prepared

    def set_spherical_joints(self, body_pairs):
        bp = np.ascontiguousarray(np.asarray(body_pairs, dtype=np.uint32).reshape(-1, 2))
        lib().orc_physics_set_joints(self.h, bp.ctypes.data_as(C.c_void_p) if len(bp) else None, len(bp))

    def prepared_body_count(self):
        return lib().orc_physics_prepared_body_count(self.h)

    def contact_order(self):
        ids = np.zeros(self.n_prepared, dtype=np.uint64)
        lib().orc_physics_contact_order(self.h, _p(ids))
        return ids

    def accumulated_impulses(self):
        out = np.zeros((self.n_prepared, 3), dtype=np.float32)
        lib().orc_physics_accumulated_impulses(self.h, _p(out))
        return out

    def advance_momenta(self, dt):
        lib().orc_physics_advance_momenta(self.h, dt)

    def solve(self):
        lib().orc_physics_solve(self.h)

    def advance_configurations(self, dt):
        lib().orc_physics_advance_configurations(self.h, dt)

    def step(self, contacts, dt):
        c = np.ascontiguousarray(contacts)
        self.n_prepared = lib().orc_physics_step(self.h, _p(c), len(c), dt)
        return self.n_prepared
