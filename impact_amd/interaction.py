"""Host-side mirror of the reference's voxel object interaction bookkeeping for rigid bodies (SURVEY §8 row a14,
impact_voxel/src/interaction.rs:224-602), above the C ABI: what happens to the rigid body of a voxel object after voxels were
removed from it (edit ops, fracturing) — disconnected regions become new objects with their own bodies, the parent's body takes
the remaining inertial properties."""
import ctypes as C

import numpy as np

from . import capi
from .capi import check, ptr
from .voxel import VoxelObject


class RemovedMassFate:
    """`RemovedMassFate` (interaction.rs): whether the momentum of the removed mass stays with the object"""
    TRANSFERRED = 0
    DESTROYED = 1


def apply_updated_inertial_properties_to_rigid_body(rigid_body: np.ndarray, moments64, original_local_center_of_mass, preserving_momentum=False):
    """`apply_updated_inertial_properties_to_rigid_body[_preserving_momentum]` (interaction.rs:405-487); `rigid_body` is one
    RIGID_BODY_DTYPE record, updated in place; returns the new local centre of mass"""
    b = np.ascontiguousarray(rigid_body, dtype=capi.RIGID_BODY_DTYPE).reshape(1).copy()
    m = np.ascontiguousarray(moments64, dtype=np.float64)
    com = np.zeros(3, dtype=np.float32)
    check(capi.lib().ivx_apply_updated_inertial_properties(ptr(b), ptr(m), ptr(np.ascontiguousarray(original_local_center_of_mass, dtype=np.float32)),
                                                           1 if preserving_momentum else 0, ptr(com)))
    return b[0], com


def offset_reference_point(moments64, offset):
    """`VoxelObjectInertialPropertyManager::offset_reference_point_by` (object/inertia.rs:257-267)"""
    m = np.ascontiguousarray(moments64, dtype=np.float64).copy()
    check(capi.lib().ivx_offset_reference_point(ptr(m), ptr(np.ascontiguousarray(offset, dtype=np.float32))))
    return m


def determine_extracted_voxel_object_dynamics(moments64_in_parent_frame, origin_offset_in_parent, voxel_extent, original_local_center_of_mass,
                                              parent_rigid_body):
    """`determine_extracted_voxel_object_dynamics` (interaction.rs:503-585) -> (fragment body record, moments about the fragment's own
    grid origin, its local centre of mass)"""
    m = np.ascontiguousarray(moments64_in_parent_frame, dtype=np.float64).copy()
    pb = np.ascontiguousarray(parent_rigid_body, dtype=capi.RIGID_BODY_DTYPE).reshape(1)
    fb = np.zeros(1, dtype=capi.RIGID_BODY_DTYPE)
    com = np.zeros(3, dtype=np.float32)
    check(capi.lib().ivx_extracted_object_dynamics(ptr(m), ptr(np.ascontiguousarray(origin_offset_in_parent, dtype=np.uint32)), voxel_extent,
                                                   ptr(np.ascontiguousarray(original_local_center_of_mass, dtype=np.float32)), ptr(pb), ptr(fb), ptr(com)))
    return fb[0], m, com


def handle_voxel_object_after_removing_voxels(voxel_object: VoxelObject, densities, moments64, rigid_body, original_local_center_of_mass,
                                              removed_mass_fate=RemovedMassFate.TRANSFERRED, capacity=64):
    """`handle_voxel_object_after_removing_voxels` (interaction.rs:224-403) without the anchors -> dict(original_object_empty,
    rigid_body (updated record), moments64 (the parent's manager afterwards), new_local_center_of_mass, extracted: list of dicts with
    voxel_object, origin_offset_in_parent, rigid_body, moments64, local_center_of_mass)"""
    d = np.zeros(256, dtype=np.float32)
    src = np.asarray(densities, dtype=np.float32)
    d[: src.size] = src
    m = np.ascontiguousarray(moments64, dtype=np.float64).copy()
    b = np.ascontiguousarray(rigid_body, dtype=capi.RIGID_BODY_DTYPE).reshape(1).copy()
    out = np.zeros(capacity, dtype=capi.EXTRACTED_OBJECT_DTYPE)
    n = C.c_size_t(0)
    empty = C.c_int(0)
    com = np.zeros(3, dtype=np.float32)
    check(capi.lib().ivx_handle_voxel_object_after_removing_voxels(voxel_object.h, ptr(d), ptr(m), ptr(b),
                                                                   ptr(np.ascontiguousarray(original_local_center_of_mass, dtype=np.float32)),
                                                                   int(removed_mass_fate), ptr(out), capacity, C.byref(n), C.byref(empty), ptr(com)))
    voxel_object._region_count = None
    extracted = []
    for e in out[: n.value]:
        extracted.append({"voxel_object": voxel_object._wrap_child(C.c_void_p(int(e["grid"]))), "origin_offset_in_parent": tuple(int(x) for x in e["origin_offset_in_parent"]),
                          "rigid_body": e["body"].copy(), "moments64": e["moments"].copy(), "local_center_of_mass": e["local_center_of_mass"].copy()})
    return {"original_object_empty": bool(empty.value), "rigid_body": b[0], "moments64": m, "new_local_center_of_mass": com, "extracted": extracted}
