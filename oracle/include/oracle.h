/* ORACLE — test infrastructure only.
 *
 * C interface of the CPU restatement of lars-frogner/Impact's per-frame voxel physics path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (impact_amd/, include/impact_voxel_hip.h) never links or calls it.
 *
 * Parity pinning status: the reference is Rust and cannot be built here, so the
 * oracle is pinned against the reference's own known-answer tests re-typed in tests/test_oracle_*.py
 * (adjacency patterns, occupied ranges, chunk flag bits, vertex/index material packing, inertia of
 * boxes/chunks, sphere-collision outcomes). Bit-level behaviour of the un-vendored glam 0.30.10 /
 * simdnoise 3.1.7 / fastrand 2.3.0 crates is "parity unpinned" (see DESIGN.md); so are two iteration ORDERS that come from
 * hashbrown 0.16 + rustc-hash 2.1 maps (the chunk order of the mutual voxel-object contacts and of the incremental remesh):
 * the oracle walks chunks in chunk-linear order there.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_object orc_object;
typedef struct orc_mesh orc_mesh;
typedef struct orc_probes orc_probes;

/* SDFNode (generation/sdf/atomic.rs:62-81); same 32-byte layout as ivx_sdf_node */
typedef struct {
    uint32_t kind; /* 0 sphere 1 capsule 2 box 3 translation 4 rotation 5 scaling 7 union 8 subtraction 9 intersection */
    uint32_t child1, child2, pad;
    float p[4];
} orc_sdf_node;

/* ProcessedSDFNode (atomic.rs:83-102); same 128-byte layout as ivx_sdf_processed_node */
typedef struct {
    uint32_t kind, leaf_count;
    float transform[16]; /* column-major, root -> node space */
    float domain_lo[3], domain_hi[3];
    float margin;
    float a, b, c;
    uint32_t reserved[4];
} orc_sdf_processed_node;

/* per-chunk state in dense (all chunks stored) form; same 8-byte layout as ivx_chunk_info */
typedef struct {
    uint8_t kind;     /* 0 void, 1 uniform, 2 non-uniform (after derived state) */
    uint8_t gen_kind; /* kind right after generation */
    uint8_t flags;    /* VoxelChunkFlags (object.rs:163-188); 0 for void/uniform */
    uint8_t uniform_type;
    uint16_t face_dist; /* 2 bits per face at bit 2*(2*dim+side): 0 empty 1 full 2 mixed */
    uint8_t region_count, boundary_region_count;
} orc_chunk_info;

/* generators ------------------------------------------------------------------------------- */
int orc_sdf_compile(const orc_sdf_node* nodes, int n, uint32_t root, orc_sdf_processed_node* out, int cap,
                    float domain[6], int* stack_size);
orc_object* orc_object_from_sdf(const orc_sdf_node* nodes, int n, uint32_t root, float voxel_extent, uint8_t voxel_type);
/* OffsetBoxVoxelGenerator (object.rs:3387-3504), voxel extent 0.25 */
orc_object* orc_object_from_box(const int shape[3], const int offset[3], uint8_t type, int8_t sd, uint8_t flags);
/* ManualVoxelGenerator<N> (object.rs:3393-3561): cells[n*n*n] (i,j,k) order, non-zero = maximally inside */
orc_object* orc_object_from_manual(int n, const uint8_t* cells, const int offset[3]);
/* dense chunk-tiled planes; emptiness = sd >= 0, void = sd > 100 (generation.rs:336-355) */
orc_object* orc_object_from_dense(const int chunk_counts[3], float voxel_extent, const int8_t* sdf, const uint8_t* type);
void orc_object_free(orc_object*);

void orc_update_occupied_voxel_ranges(orc_object*);
void orc_compute_all_derived_state(orc_object*);
/* out: cc[3], stored non-uniform chunk count, occ_chunk lo/hi x3, occ_voxel lo/hi x3, grid shape[3] */
void orc_object_info(const orc_object*, int32_t out[19]);
float orc_object_extent(const orc_object*);
/* dense chunk-tiled export: arrays of n_chunks*4096 (index = chunk*4096 + (i<<8|j<<4|k)) */
void orc_export_dense(const orc_object*, int8_t* sdf, uint8_t* type, uint8_t* flags, uint8_t* local_labels, orc_chunk_info* info);
/* reference (sparse) layout: data_offset per chunk (-1 if not stored), AoS voxels (type,sd,flags) */
void orc_export_sparse(const orc_object*, int32_t* data_offsets, uint8_t* voxels_aos3, uint8_t* labels);

/* mesh ------------------------------------------------------------------------------------- */
orc_mesh* orc_mesh_recreate(const orc_object*);
/* all-cores variants (OpenMP over chunks; bench.py's cpu_baseline_all_cores): the object comes back with occupied ranges and derived
 * state computed; results are identical to the sequential entry points (tests/test_oracle_parallel.py) */
orc_object* orc_object_from_sdf_parallel(const orc_sdf_node* nodes, int n, uint32_t root, float voxel_extent, uint8_t voxel_type, int threads);
orc_mesh* orc_mesh_recreate_parallel(const orc_object*, int threads);
void orc_inertia_parallel(const orc_object*, const float densities[256], float out32[10], int threads);
/* VoxelObjectMesh::sync_with_voxel_object (mesh.rs:355-456): re-mesh the invalidated chunks (one byte per chunk), reusing freed buffer ranges
 * through the ChunkSubmeshManager (mesh.rs:699-849) and its RangeAllocators; chunks are visited in chunk-linear order (the reference's hash-set
 * order is unpinned). Buffers only grow; submeshes keep the manager's slot order. */
void orc_mesh_sync(orc_mesh*, const orc_object*, const uint8_t* invalidated_chunks);
int orc_mesh_modifications(const orc_mesh*, uint32_t* ranges4, int cap, int* chunks_were_removed); /* mesh.rs:826-836 */
void orc_mesh_report_synchronized(orc_mesh*);                                                    /* mesh.rs:838-841 */
void orc_range_allocator_script(const int64_t* ops, int n_ops, int64_t* results);
void orc_mesh_counts(const orc_mesh*, uint32_t out[3]); /* vertices, indices, submeshes */
/* submeshes: 16 u32 each = chunk[3], index_offset, index_count, obscured[8], vertex_offset, vertex_count, 0 */
void orc_mesh_get(const orc_mesh*, float* positions, float* normals, uint32_t* indices, uint8_t* index_materials, uint32_t* submeshes);
void orc_mesh_free(orc_mesh*);
/* padded 18^3 chunk SDF (object/sdf.rs:181-508); returns 0 if the chunk is not an exposed non-uniform chunk */
int orc_chunk_sdf(const orc_object*, int ci, int cj, int ck, float* values5832, uint8_t* types5832);
void orc_vertex_materials(const uint8_t has_voxel[8], const uint8_t materials[8], uint8_t out_indices[8], uint8_t out_weights[8]);
void orc_index_materials(const uint8_t vm_indices[24], const uint8_t vm_weights[24], uint8_t out[24]);

/* inertia ---------------------------------------------------------------------------------- */
void orc_inertia(const orc_object*, const float densities[256], float out32[10], double out64[10]);
void orc_derive_inertial_properties(const float moments[10], float out[22]);

/* connected regions -------------------------------------------------------------------------- */
/* labels: dense x-major (nx,ny,nz) u32, 0xFFFFFFFF for empty; returns component count; labels may be NULL */
uint32_t orc_region_labels(const orc_object*, uint32_t* labels);

/* rigid bodies + sequential-impulses contact solver (impact_physics) ----------------------------- */
typedef struct orc_physics orc_physics;
/* DynamicRigidBody, #[repr(C)] 152 bytes (impact_physics/src/rigid_body.rs:94-103); matrices column-major,
 * orientation = (x, y, z, w) */
typedef struct {
    float mass;
    float inertia[9], inv_inertia[9];
    float position[3];
    float orientation[4];
    float momentum[3], angular_momentum[3];
    float total_force[3], total_torque[3];
} orc_rigid_body;
/* KinematicRigidBody, 56 bytes (rigid_body.rs:108-117); angular velocity = unit axis + speed */
typedef struct {
    float position[3];
    float orientation[4];
    float velocity[3];
    float angular_axis[3];
    float angular_speed;
} orc_kinematic_body;
/* ContactWithID flattened (constraint/contact.rs:23-57) + the two rigid bodies; body index bit 31 =
 * kinematic body. flags bit 0: first contact of a manifold (one collision). 64 bytes. */
typedef struct {
    uint64_t id;
    uint32_t body_a, body_b;
    float position[3];
    float normal[3];
    float depth;
    float restitution, static_friction, dynamic_friction;
    uint32_t flags, pad;
} orc_contact;
/* ConstraintSolverConfig (constraint/solver.rs:41-57, 374-384) */
typedef struct {
    uint32_t n_iterations;
    float old_impulse_weight;
    uint32_t n_positional_correction_iterations;
    float positional_correction_factor;
} orc_solver_config;

orc_physics* orc_physics_create(void);
void orc_physics_free(orc_physics*);
void orc_physics_set_config(orc_physics*, const orc_solver_config*);
/* a14: rigid bodies after voxel removal (impact_voxel/src/interaction.rs:405-602). moments = 10 f32 of a VoxelObjectInertialPropertyManager:
 * mass, first moments, moments of inertia, products of inertia (xy, yz, zx) about the grid origin */
void orc_offset_reference_point(float moments[10], const float offset[3]); /* object/inertia.rs:257-267 */
void orc_apply_updated_inertial_properties(orc_rigid_body* body, const float moments[10], const float original_local_com[3], int preserve_momentum,
                                           float new_local_com[3]);
void orc_extracted_object_dynamics(float moments[10], const int origin_offset_in_parent[3], float voxel_extent, const float original_local_com[3],
                                   const orc_rigid_body* parent, orc_rigid_body* fragment, float new_local_com[3]);

void orc_physics_set_bodies(orc_physics*, const orc_rigid_body* dyn, int n_dyn, const orc_kinematic_body* kin, int n_kin);
void orc_physics_get_bodies(const orc_physics*, orc_rigid_body* dyn, orc_kinematic_body* kin);
void orc_rigid_body_new(orc_rigid_body* out, float mass, const float inertia[9], const float inv_inertia[9], const float position[3],
                        const float orientation[4], const float velocity[3], const float angular_velocity[3]);
void orc_rigid_body_motion(const orc_rigid_body*, float velocity[3], float angular_velocity[3]);
int orc_sphere_sphere_contact(const float ca[3], float ra, const float cb[3], float rb, float position[3], float normal[3], float* depth);
int orc_sphere_plane_contact(const float c[3], float r, const float plane_normal[3], float plane_displacement, float position[3],
                             float normal[3], float* depth);
int orc_physics_prepare(orc_physics*, const orc_contact*, int n);
void orc_physics_set_joints(orc_physics*, const uint32_t* body_pairs, int n_joints); /* SphericalJoint: a placeholder in the reference (no impulses); its bodies become constrained bodies */
int orc_physics_prepared_body_count(const orc_physics*);
void orc_physics_contact_order(const orc_physics*, uint64_t* ids);
void orc_physics_accumulated_impulses(const orc_physics*, float* out3n);
void orc_physics_advance_momenta(orc_physics*, float dt);
void orc_physics_solve(orc_physics*);
void orc_physics_advance_configurations(orc_physics*, float dt);
int orc_physics_step(orc_physics*, const orc_contact*, int n, float dt);

/* split the smaller of the first two disconnected regions off (object/extraction.rs:78-596, 1901-2123) */
int orc_split_off_smallest_region(orc_object* parent, orc_object** child, int origin_offset_in_parent[3]);

/* extract_polyhedron (mode 0) / copy_polyhedron (mode 1) (object/extraction.rs:604-1768): planes4 = n x (unit normal xyz,
 * displacement), aabb = lower xyz + upper xyz, both in voxel units relative to the grid corner */
int orc_clip_polyhedron(orc_object* parent, const float* planes4, int n_planes, const float aabb[6], int mode, orc_object** child,
                        int origin_offset_in_parent[3]);
/* an absorbing sphere eats into the object (interaction/absorption.rs:801-844 over object/intersection.rs:273-395): sphere given
 * in the object's normalized space (voxel units, grid corner at the origin). Returns the number of chunks that became void. */
int orc_absorb_sphere(orc_object* o, const float center[3], float influence_radius, float sphere_radius, const float densities[256],
                      double removed64[10], uint32_t emptied_by_type[256], uint8_t* invalidated_chunks, uint32_t* touched_chunks);
/* two objects absorbing each other where they overlap (apply_mutual_absorption, interaction/absorption.rs:891-1079): every voxel of A's padded
 * overlap ranges gets sdf_subtraction(sd, max(sd, B's SDF there)), every voxel of B's overlap ranges the same against A's SDF as it was before
 * the call; rotation/translation = world -> object. stats: emptied voxels, touched chunks, removed chunks of A, then of B. */
void orc_absorb_mutual(orc_object* a, const float rotation_a[4], const float translation_a[3], const float densities_a[256], orc_object* b,
                       const float rotation_b[4], const float translation_b[3], const float densities_b[256], float smoothness, double removed_a[10],
                       double removed_b[10], uint8_t* invalidated_a, uint8_t* invalidated_b, uint64_t stats[6]);
/* the same for an absorbing capsule (interaction/absorption.rs:846-889 over object/intersection.rs:397-530) */
int orc_absorb_capsule(orc_object* o, const float segment_start[3], const float segment_vector[3], float influence_radius, float capsule_radius,
                       const float densities[256], double removed64[10], uint32_t emptied_by_type[256], uint8_t* invalidated_chunks, uint32_t* touched_chunks);
/* contacts between a sphere collidable (world space) and the surface voxels of the object (impact_voxel/src/collidable.rs:1098-1127);
 * transform_to_object_space = rotation (xyzw) then translation. Returns the number of contacts; fills at most `cap`. */
int orc_sphere_voxel_object_contacts(const orc_object* o, const float rotation_xyzw[4], const float translation[3], const float center[3], float radius,
                                     int cap, int32_t* indices, float* position, float* normal, float* depth);
/* collision probes of a meshed voxel object (VoxelObjectCollisionProbes::recompute_for_all_chunks, impact_voxel/src/collidable.rs:361-731): per
 * chunk with a submesh, the mesh vertex of the most convex curvature in every block of 8^3 / 4^3 / 2^3 / 1 voxels. submeshes: 16 u32 each as in
 * orc_mesh_get. chunk_entries: 5 u32 per chunk that got probes (ci, cj, ck, first point, end point), room for n_submeshes entries. Returns
 * the number of points; fills at most `cap`. */
int orc_collision_probes(const orc_object* o, const float* positions, const float* normals, const uint32_t* indices, const uint32_t* submeshes,
                         uint32_t n_submeshes, float* points, uint32_t cap, uint32_t* chunk_entries, uint32_t* n_entries);
/* the probes as a set that follows an incrementally re-meshed object: recompute_for_all_chunks, then sync_with_voxel_object_and_mesh after each
 * orc_mesh_sync with the same invalidated chunks (collidable.rs:394-433, 524-612; point ranges from a RangeAllocator, chunk-linear visiting order).
 * orc_probes_get: the whole point buffer (freed ranges included) and the live entries sorted by range start. */
orc_probes* orc_probes_recompute(const orc_object*, const orc_mesh*);
void orc_probes_sync(orc_probes*, const orc_object*, const orc_mesh*, const uint8_t* invalidated_chunks);
uint32_t orc_probes_get(const orc_probes*, float* points, uint32_t cap_points, uint32_t* entries, uint32_t* n_entries);
void orc_probes_free(orc_probes*);
/* contacts between two voxel objects (for_each_mutual_voxel_object_contact, impact_voxel/src/collidable.rs:859-1049): the probes of A sampled
 * against the SDF of B, then the probes of B against A. com_* = centre of mass of the object in its own space (derive_center_of_mass),
 * rotation/translation = transform_to_object_space (world -> object). which_ijk: 4 i32 per contact — 0 (A's probe) / 1 (B's probe), then the
 * probing object's voxel indices (hashed into the ContactID as [0, i, j, k]). Chunks are walked in submesh order (the reference's hash-map
 * order is unpinned, see orc_collide.cpp). Returns the number of contacts; fills at most `cap`. */
int orc_mutual_voxel_object_contacts(const orc_object* a, const float* probes_a, const uint32_t* entries_a, uint32_t n_entries_a, const float com_a[3],
                                     const float rotation_a[4], const float translation_a[3], const orc_object* b, const float* probes_b,
                                     const uint32_t* entries_b, uint32_t n_entries_b, const float com_b[3], const float rotation_b[4],
                                     const float translation_b[3], int cap, int32_t* which_ijk, float* position, float* normal, float* depth);
/* contacts between the surface voxels of the object and a capsule collidable (world space) (impact_voxel/src/collidable.rs:1257-1286) */
int orc_capsule_voxel_object_contacts(const orc_object* o, const float rotation_xyzw[4], const float translation[3], const float segment_start[3],
                                      const float segment_vector[3], float radius, int cap, int32_t* indices, float* position, float* normal,
                                      float* depth);
/* contacts between the Corner voxels of the object and a plane collidable (world space) (impact_voxel/src/collidable.rs:1176-1208) */
int orc_plane_voxel_object_contacts(const orc_object* o, const float rotation_xyzw[4], const float translation[3], const float plane_normal[3],
                                    float plane_displacement, int cap, int32_t* indices, float* position, float* normal, float* depth);

/* quantisation helpers (lib.rs:197-222) */
int8_t orc_sd_from_f32(float v);
float orc_sd_to_f32(int8_t e);

#ifdef __cplusplus
}
#endif
#endif
