"""ctypes binding of libimpact_voxel_hip.so — the exact C ABI declared in include/impact_voxel_hip.h.

This is the same surface a Rust `define_lib!` binding would see. There is no CPU
fallback anywhere in this package: if the shared library is missing or no gfx950 device is present,
calls raise `IvxError`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IMPACT_VOXEL_HIP_LIB", os.path.join(_HERE, "lib", "libimpact_voxel_hip.so"))

IVX_OK, IVX_ERR_INVALID, IVX_ERR_HIP, IVX_ERR_CAPACITY, IVX_ERR_STATE = 0, 1, 2, 3, 4

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
SUBMESH_DTYPE = np.dtype(
    [
        ("chunk_indices", "<u4", (3,)),
        ("index_offset", "<u4"),
        ("index_count", "<u4"),
        ("is_obscured_from_direction", "<u4", (2, 2, 2)),
        ("vertex_offset", "<u4"),
        ("vertex_count", "<u4"),
        ("reserved", "<u4"),
    ]
)
MOMENTS_DTYPE = np.dtype([("m64", "<f8", (10,)), ("m32", "<f4", (10,)), ("reserved", "<u4", (2,))])
REGION_DESC_DTYPE = np.dtype(
    [
        ("root_chunk", "<u4"),
        ("root_region", "<u4"),
        ("voxel_count", "<u8"),
        ("lo", "<u4", (3,)),
        ("hi", "<u4", (3,)),
        ("non_uniform_chunk_count", "<u4"),
        ("chunk_count", "<u4"),
        ("moments", "<f8", (10,)),
    ]
)
MESH_COUNTS_DTYPE = np.dtype([("n_vertices", "<u4"), ("n_indices", "<u4"), ("n_submeshes", "<u4"), ("reserved", "<u4")])
N_TIMED_STAGES = 10
IMPACT_FRACTURING_CONFIG_DTYPE = np.dtype(
    [("boundary_polar_grid_size", "<u4"), ("boundary_azimuthal_grid_size", "<u4"), ("boundary_angular_jitter", "<f4"), ("boundary_radial_jitter", "<f4"),
     ("max_fragment_count", "<u8"), ("radial_falloff_power", "<f4"), ("angular_falloff_power", "<f4"), ("radial_grid_size", "<u4"),
     ("angular_grid_size", "<u4"), ("max_position_rejections_per_sample", "<u8"), ("seed", "<u8")]
)
FRACTURING_PROPERTIES_DTYPE = np.dtype(
    [("fracturing_force", "<f4"), ("shattering_pressure", "<f4"), ("fragment_scale", "<f4"), ("min_fragment_extent", "<f4"), ("max_fragment_extent", "<f4")]
)
assert IMPACT_FRACTURING_CONFIG_DTYPE.itemsize == 56 and FRACTURING_PROPERTIES_DTYPE.itemsize == 20
SLAB_RESULT_DTYPE = np.dtype(
    [("region_count", "<u4"), ("local_region_count", "<u4"), ("first_local_component", "<u4"), ("reserved", "<u4"), ("moments", "<f8", (10,)),
     ("occupied", "<u4", (12,)), ("mesh", MESH_COUNTS_DTYPE), ("vertex_offset", "<u8"), ("index_offset", "<u8"), ("total_triangles", "<u8"),
     ("stage_ms", "<f4", (10,)), ("reserved2", "<f4", (2,))]
)
# timed slots of a step (include/impact_voxel_hip.h, IVX_N_TIMED_STAGES): the table-sized passes after the derive sweep run as roles of
# four fused launches; 6..9 are unused
STAGE_NAMES = ["sdf_sample", "derive", "post1", "post2", "emit", "assign", "unused6", "unused7", "unused8", "unused9"]
STAGE_SAMPLE, STAGE_DERIVE, STAGE_OCCUPIED, STAGE_REGIONS, STAGE_REMESH, STAGE_INERTIA, STAGE_ALL = 1, 2, 4, 8, 16, 32, 63
STEP_RESULT_DTYPE = np.dtype(
    [
        ("mesh", MESH_COUNTS_DTYPE),
        ("region_count", "<u4"),
        ("occupied", "<u4", (12,)),
        ("reserved", "<u4", (3,)),
        ("moments", MOMENTS_DTYPE),
        ("stage_ms", "<f4", (N_TIMED_STAGES,)),
        ("reserved2", "<f4", (2,)),
    ]
)
assert STEP_RESULT_DTYPE.itemsize == 256
assert CHUNK_INFO_DTYPE.itemsize == 8 and SUBMESH_DTYPE.itemsize == 64
assert MOMENTS_DTYPE.itemsize == 128 and REGION_DESC_DTYPE.itemsize == 128

# rigid bodies / contacts (impact_physics) — reference #[repr(C)] layouts
RIGID_BODY_DTYPE = np.dtype(
    [
        ("mass", "<f4"),
        ("inertia", "<f4", (9,)),       # Matrix3C, column-major
        ("inv_inertia", "<f4", (9,)),
        ("position", "<f4", (3,)),
        ("orientation", "<f4", (4,)),   # x, y, z, w
        ("momentum", "<f4", (3,)),
        ("angular_momentum", "<f4", (3,)),
        ("total_force", "<f4", (3,)),
        ("total_torque", "<f4", (3,)),
    ]
)
KINEMATIC_BODY_DTYPE = np.dtype(
    [("position", "<f4", (3,)), ("orientation", "<f4", (4,)), ("velocity", "<f4", (3,)), ("angular_axis", "<f4", (3,)), ("angular_speed", "<f4")]
)
# `ivx_collidable_query` (ivx_voxel_object_contacts_many): one collidable against one voxel object
COLLIDABLE_QUERY_DTYPE = np.dtype(
    [
        ("mode", "<i4"),
        ("rotation_xyzw", "<f4", (4,)),
        ("translation", "<f4", (3,)),
        ("shape3", "<f4", (3,)),
        ("shape3b", "<f4", (3,)),
        ("shape1", "<f4"),
        ("body_a", "<u4"),
        ("body_b", "<u4"),
        ("collidable_id_a", "<u8"),
        ("collidable_id_b", "<u8"),
        ("response", "<f4", (3,)),
        ("reserved", "<u4"),
    ],
    align=True,
)
# `ivx_mutual_query` (ivx_mutual_voxel_object_contacts_many): one pair of voxel objects
MUTUAL_QUERY_DTYPE = np.dtype(
    [
        ("a", "<u8"),
        ("b", "<u8"),
        ("rotation_a", "<f4", (4,)),
        ("translation_a", "<f4", (3,)),
        ("center_of_mass_a", "<f4", (3,)),
        ("rotation_b", "<f4", (4,)),
        ("translation_b", "<f4", (3,)),
        ("center_of_mass_b", "<f4", (3,)),
        ("collidable_id_a", "<u8"),
        ("collidable_id_b", "<u8"),
        ("body_a", "<u4"),
        ("body_b", "<u4"),
        ("response", "<f4", (3,)),
        ("reserved", "<u4"),
    ],
    align=True,
)
CONTACT_DTYPE = np.dtype(
    [
        ("id", "<u8"),
        ("body_a", "<u4"),
        ("body_b", "<u4"),
        ("position", "<f4", (3,)),
        ("normal", "<f4", (3,)),
        ("depth", "<f4"),
        ("restitution", "<f4"),
        ("static_friction", "<f4"),
        ("dynamic_friction", "<f4"),
        ("flags", "<u4"),
        ("pad", "<u4"),
    ]
)
ABSORB_RESULT_DTYPE = np.dtype([("removed_moments", "<f8", (10,)), ("emptied_voxels", "<u8"), ("touched_chunks", "<u4"), ("removed_chunks", "<u4")])
SOLVER_CONFIG_DTYPE = np.dtype(
    [("n_iterations", "<u4"), ("old_impulse_weight", "<f4"), ("n_positional_correction_iterations", "<u4"), ("positional_correction_factor", "<f4")]
)
PHYSICS_RESULT_DTYPE = np.dtype(
    [("n_contacts", "<u4"), ("n_bodies", "<u4"), ("n_levels", "<u4", (2,)), ("stage_ms", "<f4", (5,)), ("reserved", "<u4", (3,))]
)
PHYSICS_STAGE_NAMES = ["prepare", "pre_solve", "solve", "post_solve", "total"]
KINEMATIC_BIT = 0x80000000
CONTACT_MANIFOLD_START = 1
assert RIGID_BODY_DTYPE.itemsize == 152 and KINEMATIC_BODY_DTYPE.itemsize == 56 and CONTACT_DTYPE.itemsize == 64

# every symbol include/impact_voxel_hip.h declares
EXPORTED_SYMBOLS = [
    "ivx_init", "ivx_shutdown", "ivx_last_error", "ivx_synchronize", "ivx_stream",
    "ivx_grid_create", "ivx_grid_destroy", "ivx_grid_upload_dense", "ivx_grid_download_dense", "ivx_grid_device_ptr", "ivx_grid_chunk_counts", "ivx_grid_stage_counters",
    "ivx_sdf_compile", "ivx_sdf_grid_shape", "ivx_sdf_sample",
    "ivx_derive_state", "ivx_occupied_ranges",
    "ivx_remesh", "ivx_mesh_download", "ivx_mesh_device_ptr",
    "ivx_inertia",
    "ivx_label_regions", "ivx_region_labels_download", "ivx_regions_describe", "ivx_split_off_smallest_region", "ivx_split_off_all", "ivx_clip_polyhedron", "ivx_copy_polyhedra", "ivx_mesh_sync", "ivx_mesh_export", "ivx_mesh_generation", "ivx_mesh_import_open", "ivx_mesh_import_close", "ivx_mesh_modifications", "ivx_mesh_report_synchronized", "ivx_absorb_sphere", "ivx_absorb_capsule", "ivx_absorb_mutual", "ivx_absorb_sphere_enqueue", "ivx_absorb_capsule_enqueue", "ivx_absorb_collect", "ivx_grid_set_early_mesh_needs", "ivx_mesh_sync_enqueue", "ivx_mesh_sync_collect",
    "ivx_many_begin", "ivx_many_flush", "ivx_many_stats", "ivx_voxel_step_many", "ivx_absorb_sphere_many", "ivx_absorb_capsule_many", "ivx_mesh_sync_many", "ivx_offset_reference_point", "ivx_apply_updated_inertial_properties", "ivx_extracted_object_dynamics", "ivx_handle_voxel_object_after_removing_voxels", "ivx_sphere_voxel_object_contacts", "ivx_plane_voxel_object_contacts", "ivx_capsule_voxel_object_contacts", "ivx_voxel_object_contacts_many", "ivx_collision_probes_recompute", "ivx_collision_probes_sync", "ivx_collision_probes_sync_many", "ivx_collision_probes_download", "ivx_mutual_voxel_object_contacts", "ivx_mutual_voxel_object_contacts_many",
    "ivx_grid_set_sdf_program", "ivx_grid_set_densities", "ivx_voxel_step", "ivx_voxel_step_enqueue", "ivx_voxel_step_collect", "ivx_grid_set_stage_timing", "ivx_grid_set_sample_ahead",
    "ivx_halo_pack_enqueue", "ivx_halo_unpack_enqueue", "ivx_halo_pack_both_enqueue", "ivx_region_face_labels_enqueue", "ivx_region_face_pairs_enqueue",
    "ivx_step_record_words", "ivx_step_record_enqueue", "ivx_slab_remesh_enqueue",
    "ivx_halo_bytes", "ivx_halo_pack", "ivx_halo_unpack", "ivx_halo_clear",
    "ivx_region_face_bytes", "ivx_region_face_labels", "ivx_region_face_pairs",
    "ivx_world_create", "ivx_world_destroy", "ivx_world_set_bodies", "ivx_world_get_bodies", "ivx_world_set_contacts",
    "ivx_world_set_spherical_joints", "ivx_world_step", "ivx_world_step_enqueue", "ivx_world_prepare", "ivx_world_advance_momenta", "ivx_world_solve", "ivx_world_advance_configurations",
    "ivx_impact_fracturing_config_default", "ivx_generate_impact_fracture_points", "ivx_delaunay_construct", "ivx_delaunay_destroy", "ivx_delaunay_counts",
    "ivx_delaunay_download", "ivx_delaunay_aabb", "ivx_delaunay_displace_vertices", "ivx_delaunay_boundary_face_planes", "ivx_voronoi_polyhedron", "ivx_voronoi_bounded_aabb",
    "ivx_comm_unique_id", "ivx_comm_init", "ivx_comm_init_local", "ivx_comm_init_ipc", "ivx_comm_info", "ivx_comm_set_local_copies", "ivx_comm_selftest", "ivx_selftest_mesher_division", "ivx_comm_destroy", "ivx_slab_create", "ivx_slab_destroy",
    "ivx_slabs_step_enqueue", "ivx_slabs_step_collect", "ivx_slab_region_map",
    "ivx_world_set_solver_groups", "ivx_world_solver_info", "ivx_world_contact_state",
]


MESH_EXPORT_DTYPE = np.dtype([("ipc_handle", "u1", (64,)), ("dmabuf_fd", "<i4"), ("element_bytes", "<u4"), ("bytes", "<u8"), ("capacity_bytes", "<u8"),
                              ("generation", "<u8"), ("device_ptr", "<u8"), ("dmabuf_offset", "<u8"), ("dmabuf_bytes", "<u8")])

EXTRACTED_OBJECT_DTYPE = np.dtype(
    [
        ("grid", "<u8"),
        ("origin_offset_in_parent", "<u4", (3,)),
        ("reserved", "<u4"),
        ("body", RIGID_BODY_DTYPE),
        ("moments", "<f8", (10,)),
        ("local_center_of_mass", "<f4", (3,)),
        ("reserved2", "<f4"),
    ]
)
assert EXTRACTED_OBJECT_DTYPE.itemsize == 272


def extra_struct_sizes():
    """name -> (dtype, size the header documents) for the structs not asserted above"""
    return {
        "ivx_rigid_body": (RIGID_BODY_DTYPE, 152), "ivx_kinematic_body": (KINEMATIC_BODY_DTYPE, 56), "ivx_contact": (CONTACT_DTYPE, 64),
        "ivx_solver_config": (SOLVER_CONFIG_DTYPE, 16), "ivx_physics_result": (PHYSICS_RESULT_DTYPE, 48),
        "ivx_absorb_result": (ABSORB_RESULT_DTYPE, 96), "ivx_extracted_object": (EXTRACTED_OBJECT_DTYPE, 272),
    }


class IvxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ivx error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load the shared library (fails loudly when it is missing: there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch (ROCm wheel) bundles its own HIP runtime; two runtimes in one process cannot both own the GPU
    # ("No HIP GPUs are available" in whichever initialises second). Importing torch first makes this
    # library bind to the runtime torch already loaded, so device buffers and streams can be shared.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise IvxError(IVX_ERR_HIP, f"{LIB_PATH} not found — run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(LIB_PATH)
    vp, u32, i32, sz, f32 = C.c_void_p, C.c_uint32, C.c_int, C.c_size_t, C.c_float
    sig = {
        "ivx_init": (i32, [i32, vp, C.POINTER(vp)]),
        "ivx_shutdown": (None, [vp]),
        "ivx_last_error": (C.c_char_p, []),
        "ivx_synchronize": (i32, [vp]),
        "ivx_stream": (vp, [vp]),
        "ivx_grid_create": (i32, [vp, vp, f32, u32, u32, C.POINTER(vp)]),
        "ivx_grid_destroy": (None, [vp]),
        "ivx_grid_upload_dense": (i32, [vp, vp, vp, sz]),
        "ivx_grid_download_dense": (i32, [vp, vp, vp, vp, vp, vp, sz]),
        "ivx_grid_device_ptr": (vp, [vp, i32]),
        "ivx_grid_chunk_counts": (i32, [vp, vp]),
        "ivx_grid_stage_counters": (i32, [vp, vp]),
        "ivx_sdf_compile": (i32, [vp, sz, u32, vp, sz, C.POINTER(sz), vp, C.POINTER(u32)]),
        "ivx_sdf_grid_shape": (i32, [vp, vp, vp]),
        "ivx_sdf_sample": (i32, [vp, vp, sz, u32, vp, vp, C.c_uint8]),
        "ivx_derive_state": (i32, [vp]),
        "ivx_occupied_ranges": (i32, [vp, vp]),
        "ivx_remesh": (i32, [vp, vp]),
        "ivx_mesh_download": (i32, [vp, vp, vp, vp, vp, vp]),
        "ivx_mesh_device_ptr": (vp, [vp, i32]),
        "ivx_inertia": (i32, [vp, vp, vp]),
        "ivx_label_regions": (i32, [vp, C.POINTER(u32)]),
        "ivx_region_labels_download": (i32, [vp, vp, sz]),
        "ivx_regions_describe": (i32, [vp, vp, vp, sz, C.POINTER(sz)]),
        "ivx_split_off_smallest_region": (i32, [vp, C.POINTER(vp), vp, C.POINTER(i32), vp]),
        "ivx_split_off_all": (i32, [vp, C.c_size_t, vp, vp, vp, vp, vp]),
        "ivx_clip_polyhedron": (i32, [vp, vp, sz, vp, i32, C.POINTER(vp), vp, C.POINTER(i32)]),
        "ivx_copy_polyhedra": (i32, [vp, vp, vp, vp, sz, vp, vp, vp]),
        "ivx_grid_set_sdf_program": (i32, [vp, vp, sz, u32, vp, vp, C.c_uint8]),
        "ivx_grid_set_densities": (i32, [vp, vp]),
        "ivx_voxel_step": (i32, [vp, u32, vp]),
        "ivx_voxel_step_enqueue": (i32, [vp, u32]),
        "ivx_voxel_step_collect": (i32, [vp, vp]),
        "ivx_grid_set_stage_timing": (i32, [vp, u32]),
        "ivx_grid_set_sample_ahead": (i32, [vp, i32]),
        "ivx_halo_pack_enqueue": (i32, [vp, i32, vp]),
        "ivx_halo_unpack_enqueue": (i32, [vp, i32, vp]),
        "ivx_halo_pack_both_enqueue": (i32, [vp, vp, vp, i32]),
        "ivx_region_face_labels_enqueue": (i32, [vp, i32, vp]),
        "ivx_region_face_pairs_enqueue": (i32, [vp, i32, vp]),
        "ivx_step_record_words": (sz, []),
        "ivx_step_record_enqueue": (i32, [vp, vp]),
        "ivx_slab_remesh_enqueue": (i32, [vp, vp, vp]),
        "ivx_halo_bytes": (sz, [vp]),
        "ivx_halo_pack": (i32, [vp, i32, vp]),
        "ivx_halo_unpack": (i32, [vp, i32, vp]),
        "ivx_halo_clear": (i32, [vp, i32]),
        "ivx_region_face_bytes": (sz, [vp]),
        "ivx_region_face_labels": (i32, [vp, i32, vp]),
        "ivx_region_face_pairs": (i32, [vp, i32, vp, vp, sz, C.POINTER(sz)]),
        "ivx_world_create": (i32, [vp, vp, C.POINTER(vp)]),
        "ivx_world_destroy": (None, [vp]),
        "ivx_world_set_bodies": (i32, [vp, vp, sz, vp, sz]),
        "ivx_world_get_bodies": (i32, [vp, vp, vp]),
        "ivx_world_set_contacts": (i32, [vp, vp, sz, C.POINTER(sz)]),
        "ivx_mesh_sync": (i32, [vp, vp, vp]),
        "ivx_mesh_modifications": (i32, [vp, vp, sz, C.POINTER(sz), C.POINTER(i32)]),
        "ivx_mesh_report_synchronized": (i32, [vp]),
        "ivx_absorb_sphere": (i32, [vp, vp, f32, f32, vp, vp, vp, vp]),
        "ivx_absorb_sphere_enqueue": (i32, [vp, vp, f32, f32, vp]),
        "ivx_absorb_capsule_enqueue": (i32, [vp, vp, vp, f32, f32, vp]),
        "ivx_absorb_collect": (i32, [vp, vp, vp, vp]),
        "ivx_many_begin": (i32, [vp]),
        "ivx_many_flush": (i32, [vp]),
        "ivx_many_stats": (i32, [vp, vp]),
        "ivx_voxel_step_many": (i32, [vp, sz, u32, vp]),
        "ivx_absorb_sphere_many": (i32, [vp, sz, vp, vp, vp, vp, vp, vp]),
        "ivx_absorb_capsule_many": (i32, [vp, sz, vp, vp, vp, vp, vp, vp, vp]),
        "ivx_mesh_sync_many": (i32, [vp, sz, vp, vp]),
        "ivx_mesh_sync_enqueue": (i32, [vp, vp]),
        "ivx_grid_set_early_mesh_needs": (i32, [vp, i32]),
        "ivx_mesh_sync_collect": (i32, [vp, vp]),
        "ivx_offset_reference_point": (i32, [vp, vp]),
        "ivx_apply_updated_inertial_properties": (i32, [vp, vp, vp, i32, vp]),
        "ivx_extracted_object_dynamics": (i32, [vp, vp, f32, vp, vp, vp, vp]),
        "ivx_handle_voxel_object_after_removing_voxels": (i32, [vp, vp, vp, vp, vp, i32, vp, sz, C.POINTER(sz), C.POINTER(i32), vp]),
        "ivx_absorb_mutual": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp]),
        "ivx_absorb_capsule": (i32, [vp, vp, vp, f32, f32, vp, vp, vp, vp]),
        "ivx_sphere_voxel_object_contacts": (i32, [vp, vp, vp, vp, f32, C.c_uint64, C.c_uint64, u32, u32, vp, vp, sz, C.POINTER(sz)]),
        "ivx_plane_voxel_object_contacts": (i32, [vp, vp, vp, vp, f32, C.c_uint64, C.c_uint64, u32, u32, vp, vp, sz, C.POINTER(sz)]),
        "ivx_collision_probes_recompute": (i32, [vp, C.POINTER(sz)]),
        "ivx_collision_probes_sync": (i32, [vp, vp, C.POINTER(sz)]),
        "ivx_collision_probes_sync_many": (i32, [vp, C.c_size_t, vp, vp]),
        "ivx_collision_probes_download": (i32, [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)]),
        "ivx_mutual_voxel_object_contacts": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, u32, u32, vp, vp, sz, C.POINTER(sz)]),
        "ivx_mutual_voxel_object_contacts_many": (i32, [vp, C.c_size_t, vp, C.c_size_t, vp]),
        "ivx_capsule_voxel_object_contacts": (i32, [vp, vp, vp, vp, vp, f32, C.c_uint64, C.c_uint64, u32, u32, vp, vp, sz, C.POINTER(sz)]),
        "ivx_voxel_object_contacts_many": (i32, [vp, C.c_size_t, vp, vp, C.c_size_t, vp]),
        "ivx_world_step": (i32, [vp, f32, vp]),
        "ivx_world_step_enqueue": (i32, [vp, f32]),
        "ivx_world_set_spherical_joints": (i32, [vp, vp, sz]),
        "ivx_world_prepare": (i32, [vp]),
        "ivx_world_advance_momenta": (i32, [vp, f32]),
        "ivx_world_solve": (i32, [vp]),
        "ivx_world_advance_configurations": (i32, [vp, f32]),
        "ivx_world_contact_state": (i32, [vp, vp, vp, sz, C.POINTER(sz)]),
        "ivx_impact_fracturing_config_default": (None, [vp]),
        "ivx_generate_impact_fracture_points": (i32, [vp, vp, f32, vp, vp, vp, vp, vp, f32, vp, vp, sz, C.POINTER(sz), vp, sz, C.POINTER(sz)]),
        "ivx_delaunay_construct": (i32, [vp, sz, C.POINTER(vp)]),
        "ivx_delaunay_destroy": (None, [vp]),
        "ivx_delaunay_counts": (i32, [vp, vp]),
        "ivx_delaunay_download": (i32, [vp, vp, vp, vp]),
        "ivx_delaunay_aabb": (i32, [vp, vp]),
        "ivx_delaunay_displace_vertices": (i32, [vp, vp]),
        "ivx_delaunay_boundary_face_planes": (i32, [vp, vp, sz, C.POINTER(sz)]),
        "ivx_voronoi_polyhedron": (i32, [vp, u32, vp, sz, vp, sz, vp, sz, vp]),
        "ivx_voronoi_bounded_aabb": (i32, [vp, sz, vp, sz, vp, vp, C.POINTER(i32)]),
        "ivx_comm_unique_id": (i32, [vp]),
        "ivx_comm_init": (i32, [vp, i32, i32, vp, C.POINTER(vp)]),
        "ivx_comm_init_local": (i32, [vp, i32, C.POINTER(vp)]),
        "ivx_comm_init_ipc": (i32, [vp, i32, i32, C.c_char_p, C.POINTER(vp)]),
        "ivx_mesh_export": (i32, [vp, i32, vp]),
        "ivx_mesh_generation": (i32, [vp, C.POINTER(C.c_uint64)]),
        "ivx_mesh_import_open": (i32, [vp, i32, C.POINTER(vp)]),
        "ivx_mesh_import_close": (i32, [vp]),
        "ivx_comm_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "ivx_comm_set_local_copies": (i32, [vp, i32]),
        "ivx_comm_selftest": (i32, [vp]),
        "ivx_selftest_mesher_division": (i32, [vp, C.POINTER(u32)]),
        "ivx_comm_destroy": (None, [vp]),
        "ivx_slab_create": (i32, [vp, vp, i32, C.POINTER(vp)]),
        "ivx_slab_destroy": (None, [vp]),
        "ivx_slabs_step_enqueue": (i32, [vp, sz]),
        "ivx_slabs_step_collect": (i32, [vp, sz, vp]),
        "ivx_slab_region_map": (i32, [vp, i32, vp, sz, C.POINTER(sz)]),
        "ivx_world_set_solver_groups": (i32, [vp, u32]),
        "ivx_world_solver_info": (i32, [vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(code):
    if code != IVX_OK:
        raise IvxError(code, lib().ivx_last_error().decode("utf-8", "replace"))


def ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
