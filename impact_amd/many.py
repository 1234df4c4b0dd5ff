"""Many voxel objects per call (`ivx_*_many`, csrc/many.hpp): the reference's per-frame unit is the manager — every object's mesh is synced each
frame (impact_voxel/src/lib.rs:729-733), the fragments of an impact come into being together (interaction/fracturing.rs:1047-1189) — and one
small object's step, edit or sync is all launch latency. These calls run the same per-object work for N objects of one context in the launches
of one. Nothing here computes."""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import capi
from .capi import check, ptr


def _handles(objects):
    """the objects' handles as a C array of pointers (a uint64 array: one conversion for all instead of a ctypes assignment each)"""
    return np.fromiter((o.h.value if isinstance(o.h, C.c_void_p) else o.h for o in objects), dtype=np.uint64, count=len(objects))


def voxel_step_many(objects, stages: int):
    """`ivx_voxel_step_many`: `VoxelObject.step(stages)` of every object -> array of step results"""
    out = np.zeros(len(objects), dtype=capi.STEP_RESULT_DTYPE)
    check(capi.lib().ivx_voxel_step_many(ptr(_handles(objects)), len(objects), stages, ptr(out)))
    if stages & capi.STAGE_REGIONS:
        for o, n in zip(objects, out["region_count"].tolist()):
            o._region_count = n
    return out


def absorb_sphere_many(objects, centers, influence_radii, sphere_radii, densities=None, want_invalidated: bool = True):
    """`ivx_absorb_sphere_many`: one absorbing sphere per object (object-normalized coordinates) -> list of result dicts as `VoxelObject.absorb_sphere`
    returns them (without the per-type counts; `invalidated` is a uint8 view — one byte per chunk, 1 = invalidated — into one buffer for all objects)"""
    return _absorb_many(objects, centers, None, influence_radii, sphere_radii, densities, want_invalidated)


def absorb_capsule_many(objects, segment_starts, segment_vectors, influence_radii, capsule_radii, densities=None, want_invalidated: bool = True):
    """`ivx_absorb_capsule_many`: one absorbing capsule per object (segment start + segment vector, object-normalized coordinates)"""
    return _absorb_many(objects, segment_starts, segment_vectors, influence_radii, capsule_radii, densities, want_invalidated)


def _absorb_many(objects, centers, segments, influence_radii, sphere_radii, densities, want_invalidated):
    n = len(objects)
    d = np.ones(256, dtype=np.float32) if densities is None else np.ascontiguousarray(densities, dtype=np.float32)
    c = np.ascontiguousarray(np.asarray(centers, dtype=np.float32).reshape(n, 3))
    sg = None if segments is None else np.ascontiguousarray(np.asarray(segments, dtype=np.float32).reshape(n, 3))
    ri = np.ascontiguousarray(np.asarray(influence_radii, dtype=np.float32).reshape(n))
    rs = np.ascontiguousarray(np.asarray(sphere_radii, dtype=np.float32).reshape(n))
    out = np.zeros(n, dtype=capi.ABSORB_RESULT_DTYPE)
    inval = iv = None
    if want_invalidated:  # one buffer, one pointer per object into it
        sizes = np.fromiter((o.n_chunks for o in objects), dtype=np.int64, count=n)
        offs = np.concatenate(([0], np.cumsum(sizes)))
        buf = np.zeros(int(offs[-1]), dtype=np.uint8)
        iv = (buf.ctypes.data + offs[:-1]).astype(np.uint64)
        inval = [buf[offs[i]:offs[i + 1]] for i in range(n)]
    if sg is None:
        check(capi.lib().ivx_absorb_sphere_many(ptr(_handles(objects)), n, ptr(c), ptr(ri), ptr(rs), ptr(d), ptr(out), ptr(iv) if iv is not None else None))
    else:
        check(capi.lib().ivx_absorb_capsule_many(ptr(_handles(objects)), n, ptr(c), ptr(sg), ptr(ri), ptr(rs), ptr(d), ptr(out), ptr(iv) if iv is not None else None))
    for o in objects:
        o._region_count = None
    # (columns converted once: a structured-array field access per object costs more than the call's share of that object)
    rm, ev, tc, rc = out["removed_moments"], out["emptied_voxels"].tolist(), out["touched_chunks"].tolist(), out["removed_chunks"].tolist()
    return [{"removed_moments": rm[i], "emptied_voxels": ev[i], "invalidated": None if inval is None else inval[i], "touched_chunks": tc[i], "removed_chunks": rc[i]}
            for i in range(n)]


def mesh_sync_many(meshes, invalidated):
    """`ivx_mesh_sync_many`: `VoxelObjectMesh.sync_with_voxel_object` of every mesh with its own invalidated set (one byte or bool per chunk)"""
    n = len(meshes)
    objs = [m.object for m in meshes]
    inv = [a if (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]) else np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.uint8)
           for a in invalidated]
    iv = np.fromiter((a.__array_interface__["data"][0] for a in inv), dtype=np.uint64, count=n)  # (`.ctypes.data` builds a ctypes object per array)
    for a, o in zip(inv, objs):
        assert a.size == o.n_chunks
    out = np.zeros(n, dtype=capi.MESH_COUNTS_DTYPE)
    check(capi.lib().ivx_mesh_sync_many(ptr(_handles(objs)), n, ptr(iv), ptr(out)))
    for i, m in enumerate(meshes):
        m.counts = out[i]
    return meshes


def collision_probes_sync_many(objects, invalidated):
    """`ivx_collision_probes_sync_many`: `VoxelObject.collision_probes_sync` of every object with its own invalidated set (the sets its mesh was
    synced with) -> the lengths of the point buffers"""
    n = len(objects)
    inv = [a if (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]) else np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.uint8)
           for a in invalidated]
    for a, o in zip(inv, objects):
        assert a.size == o.n_chunks
    iv = np.fromiter((a.__array_interface__["data"][0] for a in inv), dtype=np.uint64, count=n)
    out = np.zeros(n, dtype=np.uint64)
    assert np.dtype(np.uintp).itemsize == 8
    check(capi.lib().ivx_collision_probes_sync_many(ptr(_handles(objects)) if n else None, n, ptr(iv) if n else None, ptr(out)))
    return out


def collidable_queries(n: int) -> np.ndarray:
    """`n` zeroed `ivx_collidable_query` records (mode 0 sphere: shape3 centre, shape1 radius; 1 plane: shape3 unit normal, shape1 displacement;
    2 capsule: shape3 segment start, shape3b segment vector, shape1 radius; rotation / translation: the object's transform_to_object_space)"""
    q = np.zeros(n, dtype=capi.COLLIDABLE_QUERY_DTYPE)
    q["rotation_xyzw"][:, 3] = 1.0
    return q


_contact_buffers = threading.local()


def _contact_buffer(capacity: int) -> np.ndarray:
    """the list a contact call writes into, kept between calls (a fresh 16 MB array per call costs more than the call: the C side fills the
    first n records, the caller gets a copy of those). Per thread: ctypes releases the GIL during the call."""
    bufs = _contact_buffers.__dict__.setdefault("by_capacity", {})
    buf = bufs.get(capacity)
    if buf is None:
        buf = bufs[capacity] = np.empty(capacity, dtype=capi.CONTACT_DTYPE)
    return buf


def voxel_object_contacts_many(objects, queries, capacity: int = 1 << 18):
    """`ivx_voxel_object_contacts_many`: collidable i against object i for all objects in the launches of one -> (contacts, offsets): object i's
    contacts are contacts[offsets[i]:offsets[i + 1]], as the single-object `sphere_contacts` / `plane_contacts` / `capsule_contacts` return them"""
    n = len(objects)
    assert len(queries) == n and queries.dtype == capi.COLLIDABLE_QUERY_DTYPE
    out = _contact_buffer(capacity)
    offsets = np.zeros(n + 1, dtype=np.uint32)
    q = np.ascontiguousarray(queries)
    check(capi.lib().ivx_voxel_object_contacts_many(ptr(_handles(objects)) if n else None, n, ptr(q) if n else None, ptr(out), capacity, ptr(offsets)))
    return out[: int(offsets[n])].copy(), offsets


def mutual_queries(pairs) -> np.ndarray:
    """the `ivx_mutual_query` array of a list of pairs, each a dict with a, b (VoxelObject), rotation_a/b (xyzw), translation_a/b (world -> object),
    center_of_mass_a/b (object space), collidable_id_a/b, body_a/b, response (restitution, static, dynamic friction) — the arguments of
    `VoxelObject.mutual_contacts`"""
    q = np.zeros(len(pairs), dtype=capi.MUTUAL_QUERY_DTYPE)
    for i, p in enumerate(pairs):
        q[i]["a"], q[i]["b"] = p["a"].h.value, p["b"].h.value
        for k in ("rotation_a", "translation_a", "center_of_mass_a", "rotation_b", "translation_b", "center_of_mass_b", "collidable_id_a", "collidable_id_b",
                  "body_a", "body_b"):
            q[i][k] = p[k]
        q[i]["response"] = p.get("response", (0.0, 0.0, 0.0))
    return q


def mutual_voxel_object_contacts_many(queries, capacity: int = 1 << 18):
    """`ivx_mutual_voxel_object_contacts_many`: the mutual contacts of every pair in merged launches -> (contacts, offsets): pair i's manifold is
    contacts[offsets[i]:offsets[i + 1]], as `VoxelObject.mutual_contacts` returns it"""
    n = len(queries)
    assert queries.dtype == capi.MUTUAL_QUERY_DTYPE
    out = _contact_buffer(capacity)
    offsets = np.zeros(n + 1, dtype=np.uint32)
    q = np.ascontiguousarray(queries)
    check(capi.lib().ivx_mutual_voxel_object_contacts_many(ptr(q) if n else None, n, ptr(out), capacity, ptr(offsets)))
    return out[: int(offsets[n])].copy(), offsets
