/* impact_voxel_hip.h — C ABI of libimpact_voxel_hip.so
 *
 * MI355X (gfx950) implementation of the per-frame deformable-voxel physics step of
 * lars-frogner/Impact, exposed as a flat C ABI so that the reference's own Rust loader
 * `define_lib!` (interop/dynamic_lib/src/macros.rs:17-183) can bind it: FFI-primitive arguments
 * only, complex data as caller-allocated (ptr, len) buffers in the little-endian layouts documented
 * here, no heap ownership crosses the boundary, errors are status codes (0 = ok) with a message
 * available from ivx_last_error(). See INTEGRATION.md for the Rust-side binding.
 *
 * Every entry point cites the reference interface it stands in for (paths relative to
 * engine/crates/). There is NO CPU fallback: every compute entry point runs HIP kernels and fails
 * with IVX_ERR_HIP if no gfx950 device is usable.
 *
 * Device data layout ("dense chunk-tiled SoA"): the object's chunk grid (cx,cy,cz) is stored
 * completely (void and uniform chunks included); chunk c = (ci*cy + cj)*cz + ck owns 4096 bytes in
 * each plane at offset c*4096, voxel (i,j,k) at (i<<8)|(j<<4)|k — the reference's own chunk and
 * voxel index math (impact_voxel/src/object.rs:3140-3199). Planes: sdf (i8 quantised signed
 * distance, impact_voxel/src/lib.rs:154-222), type (u8 VoxelType), flags (u8 VoxelFlags,
 * lib.rs:75-101), local region label (u8, object/split_detection.rs:125,154).
 */
#ifndef IMPACT_VOXEL_HIP_H
#define IMPACT_VOXEL_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IVX_OK 0
#define IVX_ERR_INVALID 1   /* bad argument / shape mismatch */
#define IVX_ERR_HIP 2       /* HIP runtime error or no usable device */
#define IVX_ERR_CAPACITY 3  /* caller buffer too small / internal limit exceeded */
#define IVX_ERR_STATE 4     /* call order violated (e.g. mesh download before remesh) */

typedef struct ivx_ctx ivx_ctx;
typedef struct ivx_grid ivx_grid;

/* SDFNode — impact_voxel/src/generation/sdf/atomic.rs:62-181. 32 bytes.
 * kind: 0 sphere{p0=radius} 1 capsule{p0=segment_length,p1=radius} 2 box{p0..2=extents}
 *       3 translation{child1,p0..2} 4 rotation{child1,p=quat xyzw} 5 scaling{child1,p0}
 *       7 union 8 subtraction 9 intersection {child1,child2,p0=smoothness}
 * (6 = multifractal noise is not supported: simdnoise is an un-vendored dependency) */
typedef struct {
    uint32_t kind, child1, child2, pad;
    float p[4];
} ivx_sdf_node;

/* ProcessedSDFNode — atomic.rs:83-102 (+ constructor-derived parameters). 128 bytes. */
typedef struct {
    uint32_t kind, leaf_count;
    float transform[16]; /* column-major root->node space */
    float domain_lo[3], domain_hi[3]; /* domain_with_margin */
    float margin;
    float a, b, c; /* sphere a=r | capsule a=half_segment b=r | box half extents | scaling a=s | binary a=smoothness b=0.25/a */
    uint32_t reserved[4];
} ivx_sdf_processed_node;

/* Per-chunk state (VoxelChunk / NonUniformVoxelChunk, object.rs:95-126,163-188; split_detection.rs:82-88). 8 bytes. */
typedef struct {
    uint8_t kind;      /* 0 void, 1 uniform, 2 non-uniform — after ivx_derive_state */
    uint8_t gen_kind;  /* kind right after generation/upload (object.rs:1890-1964) */
    uint8_t flags;     /* VoxelChunkFlags bits 0-5 IS_OBSCURED_{X,Y,Z}_DN,{X,Y,Z}_UP, bit 6 HAS_ONLY_EMPTY_VOXELS; 0 for void/uniform */
    uint8_t uniform_type;
    uint16_t face_dist; /* FaceVoxelDistribution, 2 bits per face at bit 2*(2*dim+side): 0 empty 1 full 2 mixed */
    uint8_t region_count, boundary_region_count; /* after ivx_label_regions */
} ivx_chunk_info;

/* ChunkSubmesh (impact_voxel/src/mesh.rs:94-103) + the chunk's vertex range (mesh.rs:145). 64 bytes. */
typedef struct {
    uint32_t chunk_indices[3];
    uint32_t index_offset, index_count;
    uint32_t is_obscured_from_direction[2][2][2];
    uint32_t vertex_offset, vertex_count, reserved;
} ivx_submesh;

typedef struct {
    uint32_t n_vertices, n_indices, n_submeshes, reserved;
} ivx_mesh_counts;

/* VoxelObjectInertialPropertyManager (impact_voxel/src/object/inertia.rs:20-25): mass, first moments,
 * moments of inertia (xx,yy,zz), products (xy,yz,zx) about the grid origin. f64 is what the kernels
 * accumulate in; f32 is the value rounded for the reference's struct. */
typedef struct {
    double m64[10];
    float m32[10];
    uint32_t reserved[2];
} ivx_moments;

/* One connected region of the object (result of VoxelObject::count_regions /
 * find_two_disconnected_regions, object/split_detection.rs:193-301, plus what
 * extract_disconnected_region needs to pick and size a fragment, object/extraction.rs:121-295). */
typedef struct {
    uint32_t root_chunk, root_region; /* GlobalRegionLabel of the representative local region */
    uint64_t voxel_count;
    uint32_t lo[3], hi[3]; /* occupied voxel range [lo, hi) */
    uint32_t non_uniform_chunk_count, chunk_count;
    double moments[10];
} ivx_region_desc;

/* ---- context ------------------------------------------------------------------------------- */
/* One context per process and GPU (the engine holds the voxel / rigid-body managers behind one
 * write lock per stage, engine/src/tasks.rs:394-395,1040,1047 — calls on a ctx are serialised by
 * the caller). `stream` may be NULL (library-owned stream) or a hipStream_t to run on. */
int ivx_init(int device_id, void* stream, ivx_ctx** out);
void ivx_shutdown(ivx_ctx*);
const char* ivx_last_error(void); /* thread-local, library-owned string */
int ivx_synchronize(ivx_ctx*);
void* ivx_stream(ivx_ctx*); /* the hipStream_t kernels are launched on */

/* ---- grid (VoxelObject storage, object.rs:45-57) ---------------------------------------------- */
/* chunk_counts = ceil(grid_shape/16) (object.rs:321). x_chunk_offset / global_x_chunks describe an
 * x-slab of a larger grid for domain decomposition (0 and cx for a whole object). */
int ivx_grid_create(ivx_ctx*, const uint32_t chunk_counts[3], float voxel_extent, uint32_t x_chunk_offset,
                    uint32_t global_x_chunks, ivx_grid** out);
void ivx_grid_destroy(ivx_grid*);
/* dense chunk-tiled host planes -> device + classification of every chunk as the generator would
 * (generation.rs:336-355, object.rs:1890-1964). Replaces VoxelObject::generate_without_derived_state
 * for callers that bring their own voxels (ChunkedVoxelGenerator, generation.rs:41-67). */
int ivx_grid_upload_dense(ivx_grid*, const int8_t* sdf, const uint8_t* type, size_t n_voxels);
/* device -> host; any pointer may be NULL */
int ivx_grid_download_dense(ivx_grid*, int8_t* sdf, uint8_t* type, uint8_t* flags, uint8_t* local_labels, ivx_chunk_info* info,
                            size_t n_voxels);
int ivx_grid_chunk_counts(ivx_grid*, uint32_t out[3]);
/* diagnostics of the last step: chunks the sampler evaluated per voxel, chunks with several local regions, chunks that
 * emitted a mesh, total chunks */
int ivx_grid_stage_counters(ivx_grid*, uint32_t out[4]);
/* device pointers of the planes: 0 sdf, 1 type, 2 flags, 3 local labels, 4 chunk info, 5 global region parents. Between
 * steps the planes of Void / Uniform chunks are not kept up to date (such a chunk is its 8-byte record, as in the
 * reference's store, object.rs:96-119); asking for a plane pointer (0-3) enqueues the kernel that writes them out. */
void* ivx_grid_device_ptr(ivx_grid*, int which);

/* ---- a3: SDF sample ----------------------------------------------------------------------------- */
/* SDFGenerator::new_in (atomic.rs:228-596): host-side graph compile, no GPU work. */
int ivx_sdf_compile(const ivx_sdf_node* nodes, size_t n_nodes, uint32_t root, ivx_sdf_processed_node* out, size_t cap,
                    size_t* n_out, float domain[6], uint32_t* stack_size);
/* SDFVoxelGenerator::new (generation.rs:207-258): grid shape and shifted grid centre from the domain */
int ivx_sdf_grid_shape(const float domain[6], uint32_t grid_shape[3], float shifted_grid_center[3]);
/* VoxelObject::generate_without_derived_state with an SDFVoxelGenerator + SameVoxelTypeGenerator
 * (object.rs:267-404, generation.rs:293-371, atomic.rs:633-875). grid_shape is the GLOBAL shape. */
int ivx_sdf_sample(ivx_grid*, const ivx_sdf_processed_node* nodes, size_t n_nodes, uint32_t stack_size,
                   const uint32_t grid_shape[3], const float shifted_grid_center[3], uint8_t voxel_type);

/* ---- a4: derived state -------------------------------------------------------------------------- */
/* VoxelObject::compute_all_derived_state minus region labelling (object.rs:1136-1145): voxel adjacency
 * flags, face distributions, chunk obscuredness, uniform-chunk demotion. */
int ivx_derive_state(ivx_grid*);
/* VoxelObject::update_occupied_ranges (object.rs:1149-1198): out = chunk lo/hi x3, voxel lo/hi x3. Needs ivx_derive_state first (the ranges are reduced
 * from per-chunk boxes the derive sweep leaves). The object keeps these ranges the way the reference does: the edit, contact and probe entry points
 * use them as they were at the last update — an edit refreshes them only when it removed a chunk, split and clip always do. */
int ivx_occupied_ranges(ivx_grid*, uint32_t out[12]);

/* ---- a5-a7: remesh ------------------------------------------------------------------------------ */
/* VoxelObjectMesh::recreate (mesh.rs:286-354): padded chunk SDF + Surface Nets for every exposed chunk,
 * concatenated in chunk-linear order. Results stay on the device until downloaded. */
int ivx_remesh(ivx_grid*, ivx_mesh_counts* out);
/* positions/normals: 3 f32 per vertex; indices u32; index_materials 8 u8 per index
 * (VoxelMeshIndexMaterials, mesh.rs:77-82); any pointer may be NULL */
int ivx_mesh_download(ivx_grid*, float* positions, float* normals, uint32_t* indices, uint8_t* index_materials,
                      ivx_submesh* submeshes);
/* VoxelObjectMesh::sync_with_voxel_object (mesh.rs:355-456): re-mesh only the chunks marked in `invalidated_chunks` (one byte per chunk, e.g. what
 * ivx_absorb_* reported), placing their data through the ChunkSubmeshManager's best-fit reuse of freed buffer ranges (mesh.rs:699-849,
 * impact_containers/src/range_allocator.rs); chunks that are no longer exposed or whose mesh is empty lose their submesh (swap_remove).
 * Buffers only grow: counts.n_vertices / n_indices are the buffer lengths including freed ranges, the submesh table says what is live. The chunks
 * are visited in chunk-linear order (the reference walks a hash set, an unpinned order that decides which freed range a chunk lands in). */
int ivx_mesh_sync(ivx_grid*, const uint8_t* invalidated_chunks, ivx_mesh_counts* out);
/* ... in two halves: `_enqueue` places the invalidated chunks' meshes (their sizes come with the last edit's results when the set is that
 * edit's: no count pass, no read-back) and launches their emit pass; `_collect` waits and returns the buffer lengths.
 * invalidated_chunks = NULL: the set of the edit that is IN FLIGHT (ivx_absorb_*_enqueue without its ivx_absorb_collect yet). What that edit's
 * chunks need reaches the host a third of the way down the edit's launches (the count role rings a bell of its own); the call waits for that,
 * places the meshes while the edit's region stages still run, and puts its launches behind them:
 *   ivx_absorb_sphere_enqueue -> ivx_mesh_sync_enqueue(g, NULL) -> ivx_absorb_collect -> ivx_mesh_sync_collect
 * is the edit and its re-mesh with one idle moment for the GPU less than the four calls in their usual order (same results). The edit must
 * have been enqueued with early delivery switched on for the object (ivx_grid_set_early_mesh_needs(g, 1): off by default — the count role
 * then pays a system-scope fence per workgroup, ~4 us per edit). */
int ivx_grid_set_early_mesh_needs(ivx_grid*, int on);
int ivx_mesh_sync_enqueue(ivx_grid*, const uint8_t* invalidated_chunks);
int ivx_mesh_sync_collect(ivx_grid*, ivx_mesh_counts* out);
/* VoxelMeshModifications (mesh.rs:113-123, 826-841), the hand-off to the renderer's buffers: the vertex / index ranges ivx_mesh_sync wrote since the
 * last report, in the order written, and whether any chunk lost its submesh; ivx_mesh_report_synchronized = report_gpu_resources_synchronized. */
typedef struct {
    uint32_t vertex_start, vertex_end, index_start, index_end;
} ivx_submesh_data_ranges;
int ivx_mesh_modifications(ivx_grid*, ivx_submesh_data_ranges* out, size_t cap, size_t* n_out, int* chunks_were_removed);
int ivx_mesh_report_synchronized(ivx_grid*);
/* device pointers of the mesh buffers (hand-off to a renderer without a host round trip):
 * 0 positions, 1 normals, 2 indices, 3 index materials, 4 submeshes */
void* ivx_mesh_device_ptr(ivx_grid*, int which);
/* The same buffers as handles another process or another graphics / compute API can import, so that the renderer (gpu_resource.rs:498-530,
 * 729-907 fills wgpu vertex / index / storage buffers from exactly these arrays; mesh.rs:94-123) takes the mesh without a host round trip:
 * `ipc_handle` = hipIpcMemHandle_t of the buffer (another HIP process: hipIpcOpenMemHandle), `dmabuf_fd` = a dma-buf file descriptor of it
 * (Vulkan / wgpu external memory: VK_EXT_external_memory_dma_buf; -1 where the runtime cannot make one; the caller closes it). `bytes` = the
 * part in use (after ivx_mesh_sync: the capacity, the live ranges are scattered), `generation` changes whenever a buffer was reallocated
 * (growth in ivx_remesh / ivx_voxel_step_collect / ivx_mesh_sync): handles of an older generation name freed memory — ask
 * ivx_mesh_generation every frame (no device work) and export again when it moved. which: as ivx_mesh_device_ptr.
 * `dmabuf_offset` / `dmabuf_bytes`: where the buffer lies inside the exported dma-buf object and how many bytes of it the descriptor
 * covers — the runtime exports whole pages of the underlying allocation, so a consumer binds [dmabuf_offset, dmabuf_offset + capacity_bytes)
 * of the imported memory (VkBindBufferMemory's memoryOffset). */
typedef struct {
    uint8_t ipc_handle[64];
    int32_t dmabuf_fd;
    uint32_t element_bytes;
    uint64_t bytes, capacity_bytes, generation, device_ptr;
    uint64_t dmabuf_offset, dmabuf_bytes;
} ivx_mesh_export_info;
int ivx_mesh_export(ivx_grid*, int which, ivx_mesh_export_info* out);
int ivx_mesh_generation(ivx_grid*, uint64_t* generation);
/* the importing side for a HIP consumer in another process: handle -> device pointer valid in the calling process, and back */
int ivx_mesh_import_open(const uint8_t ipc_handle[64], int device, void** device_ptr);
int ivx_mesh_import_close(void* device_ptr);

/* ---- a8: mass / inertia ------------------------------------------------------------------------- */
/* VoxelObjectInertialPropertyManager::initialized_from (object/inertia.rs:125-136, 615-790) */
int ivx_inertia(ivx_grid*, const float densities[256], ivx_moments* out);

/* ---- a9-a11: connected regions / split ---------------------------------------------------------- */
/* update_local_connected_regions_for_all_chunks + resolve_connected_regions_between_all_chunks
 * (object/split_detection.rs:305-487); region_count = VoxelObject::count_regions (255-301) */
int ivx_label_regions(ivx_grid*, uint32_t* region_count);
/* dense u32 component id per voxel (chunk-tiled; 0xFFFFFFFF = empty), ids = rank of the component's
 * representative region in chunk scan order */
int ivx_region_labels_download(ivx_grid*, uint32_t* labels, size_t n_voxels);
/* per-region descriptors in id order (find_two_disconnected_regions = the first two) */
int ivx_regions_describe(ivx_grid*, const float densities[256], ivx_region_desc* out, size_t cap, size_t* n_out);

/* VoxelObject::extract_any_disconnected_region (object/extraction.rs:78-596, 1901-2123): if the object consists of
 * several regions, the smaller of the first two (fewest non-uniform chunks, ties by chunk count, then the second)
 * is moved into a NEW grid whose chunk grid is the region's chunk box (repacked into a single chunk when it is
 * at most 2x2x2 chunks and at most 14 voxels wide); the parent loses those voxels. Both grids come back with
 * derived state and regions recomputed. outcome: 0 = one region, nothing done; 1 = *child is the new object
 * (caller destroys it), origin_offset_in_parent in voxels; 2 = the region had fewer than 8 voxels and was removed
 * without creating an object. `moved` (optional) receives the descriptor of the removed region — voxel count, box and
 * the mass moments the reference's PropertyTransferrer would have carried over (object/inertia.rs:341-560), for the
 * densities last set with ivx_grid_set_densities / ivx_regions_describe (1.0 if never set). */
int ivx_split_off_smallest_region(ivx_grid* parent, ivx_grid** child, uint32_t origin_offset_in_parent[3], int* outcome, ivx_region_desc* moved);
/* The whole loop in one call — `while find_two_disconnected_regions { extract the smaller of the first two }` (impact_voxel/src/interaction.rs:256,
 * object/extraction.rs:255-271): the object's regions are described once, the host plays the loop over the descriptors (a region's voxel
 * count, box and chunk counts do not change when another region leaves, nor does the scan order), every region that goes is moved into a grid
 * of its own (all from one device block), and the parent and all children are re-derived together. Results in the order the loop extracts
 * them: children[k] (NULL for a discarded crumb), origins3 (3 per split-off), outcomes (1 object, 2 crumb), moved (optional descriptors);
 * *n_out = split-offs made = regions - 1 (IVX_ERR_CAPACITY when cap is smaller; nothing is changed then). Same objects, voxel for voxel, as
 * the loop over ivx_split_off_smallest_region. */
int ivx_split_off_all(ivx_grid* parent, size_t cap, ivx_grid** children, uint32_t* origins3, int* outcomes, ivx_region_desc* moved, size_t* n_out);

/* VoxelObject::extract_polyhedron / copy_polyhedron (object/extraction.rs:604-1768): the part of the object inside the convex
 * polyhedron given by `n_planes` face planes (planes4 = n x {unit normal x,y,z, displacement}, outward normals) and its AABB
 * (lower xyz, upper xyz), all in the object's normalized model space (voxel units, grid corner at the origin). copy = 0
 * removes the polyhedron from the parent (which keeps max(sdf, complement(d))), copy = 1 leaves the parent untouched.
 * Outcome and child conventions as for ivx_split_off_smallest_region; outcome 0 = the AABB misses the object. */
int ivx_clip_polyhedron(ivx_grid* parent, const float* planes4, size_t n_planes, const float aabb[6], int copy, ivx_grid** child,
                        uint32_t origin_offset_in_parent[3], int* outcome);
/* Batched COPY of several polyhedra out of one object — all fragments of one impact (FracturingProcess::execute_in_parallel,
 * fracturing.rs:1047-1189, over copy_polyhedron_with_property_computer, extraction.rs:1301-1768): `planes4` holds the plane sets one after the
 * other (plane_counts[f] planes of 4 floats each), `aabbs6` six floats per set. children[f] / origins3[3 f ..] / outcomes[f] are what
 * ivx_clip_polyhedron(copy = 1) returns for set f (outcome 0 = no overlap, 1 = object, 2 = fewer than 8 voxels: no object); the work of all
 * fragments is enqueued back to back and read with two host waits in all. */
int ivx_copy_polyhedra(ivx_grid* parent, const float* planes4, const uint32_t* plane_counts, const float* aabbs6, size_t n_sets, ivx_grid** children,
                       uint32_t* origins3, int* outcomes);

/* ---- whole voxel step (resident inputs, minimal host synchronisation) ---------------------------- */
/* The per-frame chain the engine runs for a voxel object — generate (engine/src/setup/scene/voxel.rs:33 ->
 * impact_voxel/src/setup.rs:555-579), derived state, mesh (engine/src/tasks.rs:376-399) and inertial
 * properties (setup.rs:581-616) — as ONE call over inputs that already live in HBM. `stages` selects
 * what runs; results come back in one small struct. Stage durations are measured with HIP events on
 * the context's stream (no extra synchronisation). */
#define IVX_STAGE_SAMPLE 1u
#define IVX_STAGE_DERIVE 2u
#define IVX_STAGE_OCCUPIED 4u
#define IVX_STAGE_REGIONS 8u
#define IVX_STAGE_REMESH 16u
#define IVX_STAGE_INERTIA 32u
#define IVX_STAGE_ALL 63u
/* timed slots of a step: 0 sample (k_sdf_super, k_sdf_prepass, k_sdf_eval), 1 derive (k_chunk_pre, k_derive: flags, chunk state, chunk-local
 * regions, chunk moments), 2 post1 (one launch: mesher count | region merge by chunk columns | occupied slots | moment partial sums),
 * 3 post2 (one launch: exact local numbering, then multi-region merge | mesher scan | moments and occupied ranges final), 4 emit (one launch:
 * region forest flatten | mesher emit), 5 assign (component ids), 6..9 unused */
#define IVX_N_TIMED_STAGES 10
typedef struct {
    ivx_mesh_counts mesh;
    uint32_t region_count;
    uint32_t occupied[12];
    uint32_t reserved[3];
    ivx_moments moments;
    float stage_ms[IVX_N_TIMED_STAGES];
    float reserved2[2];
} ivx_step_result;
/* ---- voxel edit ops --------------------------------------------------------------------------------------
 * apply_sphere_absorption (impact_voxel/src/interaction/absorption.rs:801-844) over modify_voxels_within_sphere
 * (object/intersection.rs:283-395): an absorbing sphere eats into the object. The sphere is given in the object's normalized space
 * (voxel units, lower grid corner at the origin): `influence_radius` = radius + 2 voxels bounds the voxels that are visited,
 * `sphere_radius` enters the new signed distance max(sd, -(|p - c| - R)). Needs current derived state and regions; leaves
 * them current (flags, chunk kinds, regions of the edited object). `removed_moments` are the ten moments (layout of
 * ivx_moments.m64) of the voxels that became empty — what VoxelObjectInertialPropertyUpdater::remove_voxel subtracts
 * (object/inertia.rs:377-394); `emptied_by_type` (256 counts, may be NULL) is what the absorbed-voxel tracker registers;
 * `invalidated_chunks` (one byte per chunk, may be NULL) is the set handle_chunk_voxels_modified registers for remeshing
 * (intersection.rs:532-598). */
typedef struct {
    double removed_moments[10];
    uint64_t emptied_voxels;
    uint32_t touched_chunks; /* chunks with a voxel centre inside the influence sphere */
    uint32_t removed_chunks; /* chunks that became Void */
} ivx_absorb_result;
int ivx_absorb_sphere(ivx_grid*, const float center[3], float influence_radius, float sphere_radius, const float densities[256],
                      ivx_absorb_result* out, uint32_t* emptied_by_type, uint8_t* invalidated_chunks);
/* apply_capsule_absorption (interaction/absorption.rs:846-889) over modify_voxels_within_capsule (object/intersection.rs:417-530):
 * the same with an absorbing capsule — segment start + vector in the object's normalized space, `influence_radius` = radius + 2
 * voxels. Per chunk the segment is clipped against the chunk box grown by the radius (Capsule::trim_segment_outside_aab,
 * impact_geometry/src/capsule.rs:144-164) and only the voxel ranges under the clipped capsule's box are visited; a voxel is inside
 * when its centre is within the radius of the whole segment, boundary included (capsule.rs:225-250). */
int ivx_absorb_capsule(ivx_grid*, const float segment_start[3], const float segment_vector[3], float influence_radius, float capsule_radius,
                       const float densities[256], ivx_absorb_result* out, uint32_t* emptied_by_type, uint8_t* invalidated_chunks);
/* The same edits in two halves, for a frame that has other work to put on the stream (or other objects to edit) before it waits:
 * `_enqueue` launches the edit, the sweep over the touched chunks and their neighbours, the region resolve and the count of what the
 * invalidated meshes need, and returns; ivx_absorb_collect waits ONCE and delivers what ivx_absorb_sphere / _capsule return. One edit per
 * object in flight; every other call on the object waits for the collect. */
int ivx_absorb_sphere_enqueue(ivx_grid*, const float center[3], float influence_radius, float sphere_radius, const float densities[256]);
int ivx_absorb_capsule_enqueue(ivx_grid*, const float segment_start[3], const float segment_vector[3], float influence_radius, float capsule_radius,
                               const float densities[256]);
int ivx_absorb_collect(ivx_grid*, ivx_absorb_result* out, uint32_t* emptied_by_type, uint8_t* invalidated_chunks);
/* apply_mutual_absorption (interaction/absorption.rs:891-1079): two objects eat into each other where they overlap. Every voxel of A's overlap
 * ranges (determine_voxel_ranges_encompassing_intersection, padded by one voxel of B) that is not maximally outside gets
 * sdf_subtraction(sd, max(sd, B's SDF at its centre), smoothness) with B's SDF sampled trilinearly (sample_voxel_object_sdf, object/sdf.rs:636-675);
 * every voxel of B's overlap ranges the same against A's SDF as it was before the call. rotation (xyzw) + translation = each object's world -> object
 * transform. Results per object as for ivx_absorb_sphere; both objects need current derived state and regions, and keep them current. */
int ivx_absorb_mutual(ivx_grid* a, const float rotation_a[4], const float translation_a[3], const float densities_a[256], ivx_grid* b,
                      const float rotation_b[4], const float translation_b[3], const float densities_b[256], float smoothness, ivx_absorb_result* out_a,
                      ivx_absorb_result* out_b, uint8_t* invalidated_chunks_a, uint8_t* invalidated_chunks_b);

/* ---- many objects per call ----------------------------------------------------------------------------------------------------------
 * The reference's per-frame unit is the manager, not the object: every voxel object's mesh is synced each frame (impact_voxel/src/lib.rs:729-733,
 * engine/src/tasks.rs:376-399), the fragments of an impact come into being together (interaction/fracturing.rs:1047-1189), its stress scene holds
 * a thousand small objects. One object's step, edit or sync is about ten launches of a few microseconds — for a small object all of its cost.
 * These calls run the same per-object work for N objects of one context in the launches of ONE: each launch of the chain is issued once for
 * all objects (csrc/many.hpp), their results arrive together, the host waits once. Same results per object as the single-object calls.
 *   ivx_voxel_step_many      ivx_voxel_step for every object (stages without IVX_STAGE_SAMPLE merge; a sample stage runs object by object)
 *   ivx_absorb_sphere_many   ivx_absorb_sphere, one sphere per object (centers3: 3 floats per object); `invalidated_chunks`: NULL, or one
 *                            array per object (entries may be NULL)
 *   ivx_absorb_capsule_many  ivx_absorb_capsule, one capsule per object (segment start and segment vector: 3 floats each per object)
 *   ivx_mesh_sync_many       ivx_mesh_sync for every object
 * A call that fails half way (an object refused, a launch failed) drains the stream and discards what the objects before it had in flight:
 * no object is left "pending"; step the objects again before using their derived state.
 * ivx_many_begin / ivx_many_flush expose the mechanism itself: between them the `_enqueue` calls of objects of the context are recorded — a
 * chain per object, in the order the objects first appear — and the flush issues them merged, front by front. A call on an object of another
 * context inside the bracket is issued on that context's stream, unrecorded. ivx_many_flush reports a failure of any flush since the begin
 * (launches that a failed flush dropped never run; `_collect` calls that wait for them fail with IVX_ERR_HIP). The recorder and its staging
 * blocks belong to the context and go with ivx_shutdown. A begin on a thread whose last bracket was never flushed (a caller that failed in
 * between) closes that bracket first: what it recorded goes out, the thread is not left refusing brackets. ivx_many_stats: out[0] launches recorded, [1] merged launches issued, [2] flushes
 * made on this context so far. */
int ivx_many_begin(ivx_ctx*);
int ivx_many_flush(ivx_ctx*);
int ivx_many_stats(ivx_ctx*, uint64_t out[3]);
int ivx_voxel_step_many(ivx_grid* const* grids, size_t n, uint32_t stages, ivx_step_result* out);
int ivx_absorb_sphere_many(ivx_grid* const* grids, size_t n, const float* centers3, const float* influence_radii, const float* sphere_radii,
                           const float densities[256], ivx_absorb_result* out, uint8_t* const* invalidated_chunks);
int ivx_absorb_capsule_many(ivx_grid* const* grids, size_t n, const float* segment_starts3, const float* segment_vectors3, const float* influence_radii,
                            const float* capsule_radii, const float densities[256], ivx_absorb_result* out, uint8_t* const* invalidated_chunks);
int ivx_mesh_sync_many(ivx_grid* const* grids, size_t n, const uint8_t* const* invalidated_chunks, ivx_mesh_counts* out);

/* make the compiled SDF program / the voxel-type densities resident on the device */
int ivx_grid_set_sdf_program(ivx_grid*, const ivx_sdf_processed_node* nodes, size_t n_nodes, uint32_t stack_size,
                             const uint32_t grid_shape[3], const float shifted_grid_center[3], uint8_t voxel_type);
int ivx_grid_set_densities(ivx_grid*, const float densities[256]);
int ivx_voxel_step(ivx_grid*, uint32_t stages, ivx_step_result* out);
/* The same in two halves: enqueue launches the kernels of `stages` and returns without waiting (several calls may follow
 * each other, e.g. the phases of the multi-GPU protocol with halo traffic in between); collect waits once, fetches the
 * results of everything enqueued since the last collect and, if the mesh outgrew the buffers kept from earlier steps,
 * grows them and repeats the emit pass. */
int ivx_voxel_step_enqueue(ivx_grid*, uint32_t stages);
int ivx_voxel_step_collect(ivx_grid*, ivx_step_result* out);
/* Stage timing (ivx_step_result::stage_ms) costs event records on the stream (about 2 us each on the GPU's queue): slot_mask bit i = time
 * slot i. Default: every slot. 0 turns timing off (stage_ms reads 0); a single bit times one slot with two records per step. */
int ivx_grid_set_stage_timing(ivx_grid*, uint32_t slot_mask);
/* Sample-ahead, for callers that sample the resident program step after step (the voxel generator of a streamed or re-generated object,
 * generation.rs:293-371 per chunk — the bench's headline step): with `on`, a sample stage also enqueues the NEXT sample stage's interval
 * pre-pass — it reads nothing but the program and the grid's geometry — on the context's second stream behind its own evaluator, and the next
 * sample stage starts at its evaluator. Same bytes either way. A step that never comes costs one unused pre-pass;
 * ivx_grid_set_sdf_program and ivx_grid_destroy wait for one that is under way. Off by default. */
int ivx_grid_set_sample_ahead(ivx_grid*, int on);

/* ---- multi-GPU: x-slab halos ----------------------------------------------------- */
/* Face planes of (sdf,type) and boundary chunk info, packed contiguously for torch.distributed /
 * RCCL send-recv: bytes = ivx_halo_bytes(). side 0 = lower x face, 1 = upper x face. The buffers are
 * device pointers owned by the caller (e.g. a torch tensor). */
size_t ivx_halo_bytes(ivx_grid*);
int ivx_halo_pack(ivx_grid*, int side, void* device_buf);
int ivx_halo_unpack(ivx_grid*, int side, const void* device_buf); /* install as ghost layer on `side` */
int ivx_halo_clear(ivx_grid*, int side);                          /* no neighbour: outside the grid */
/* stream-ordered variants (no host wait): for callers whose communication runs on the context's stream */
int ivx_halo_pack_enqueue(ivx_grid*, int side, void* device_buf);
/* installs the buffer itself as the ghost layer (no copy): it must be 16-byte aligned and stay untouched until the next
 * ivx_halo_unpack* / ivx_halo_clear of that side */
int ivx_halo_unpack_enqueue(ivx_grid*, int side, const void* device_buf);
/* both faces in one launch (either buffer may be NULL); with_face_labels != 0 appends the face planes of component ids
 * (ivx_region_face_bytes() bytes) right behind the ivx_halo_bytes() of each message */
int ivx_halo_pack_both_enqueue(ivx_grid*, void* lower_buf, void* upper_buf, int with_face_labels);

/* Cross-slab connected regions: after ivx_label_regions on every slab, exchange the face planes of
 * component ids with the x neighbours, list the distinct (own component, neighbour component) pairs
 * that touch across the face, all-gather the pairs and finish the union on the host. This carries the
 * reference's cross-chunk region connections (object/split_detection.rs:323-487, 1046-1325) across ranks.
 * ivx_region_face_labels writes cy*cz*256 u16 (0xFFFF = empty voxel; a slab holds fewer than 65 535 components) to a DEVICE buffer;
 * ivx_region_face_pairs takes the neighbour's plane (device) and returns sorted unique pairs
 * (pairs[2i] = own id, pairs[2i+1] = neighbour id) in HOST memory. */
size_t ivx_region_face_bytes(ivx_grid*);
int ivx_region_face_labels(ivx_grid*, int side, void* device_buf);
int ivx_region_face_pairs(ivx_grid*, int side, const void* neighbour_face_labels_device, uint32_t* pairs, size_t cap, size_t* n_out);
/* Stream-ordered form of the same exchange, ending in ONE fixed-size record per slab that is written on the device and can go
 * straight into an all-gather: ivx_step_record_words() 64-bit words = [0] components of the slab, [1] number of (own,
 * neighbour) pairs across the upper face, [2..14) occupied ranges, [14..17) mesh vertices / indices / submeshes, [17] error
 * flags, [18..28) the 10 moments (f64 bit patterns), [28..) the pairs (at most 4096). */
int ivx_region_face_labels_enqueue(ivx_grid*, int side, void* device_buf);
int ivx_region_face_pairs_enqueue(ivx_grid*, int side, const void* neighbour_face_labels_device);
size_t ivx_step_record_words(void);
int ivx_step_record_enqueue(ivx_grid*, void* device_record);
/* The three calls a slab's last phase makes — the pairs across its upper face (from the neighbour's face ids; null: no upper neighbour),
 * ivx_voxel_step_enqueue(IVX_STAGE_REMESH), the record — in the remesh stage's own four launches. The step's small results land in the
 * grid's host-mapped block on the way: once the stream is idle (the protocol's doorbell) ivx_voxel_step_collect returns them without a
 * launch of its own. (What ivx_slabs_step_enqueue calls) */
int ivx_slab_remesh_enqueue(ivx_grid*, const void* neighbour_face_ids_device, void* device_record);

/* ---- a13 / §8f-3: from an impact to fragment plane sets (host code: a few hundred points per impact) ------------------------------------
 * generate_impact_fracture_points (impact_voxel/src/interaction/fracturing.rs:1710-2015), DelaunayTetrahedralization
 * (impact_tesselation/src/delaunay.rs), VoronoiPolyhedron (impact_tesselation/src/voronoi.rs). The fragments of an impact are then
 *   region   = ivx_clip_polyhedron(object, ivx_delaunay_boundary_face_planes(Delaunay(boundary points)), its aabb, extract)   fracturing.rs:1537-1632
 *   cells    = for v in 4 .. vertices(Delaunay(fracture points - region origin)): ivx_voronoi_polyhedron(v), ivx_voronoi_bounded_aabb
 *   fragments= ivx_copy_polyhedra(region, planes displaced by -0.1, boxes)                                                   fracturing.rs:1190-1240
 * Vertex indices: 0..3 are the ad-hoc bounding tetrahedron, the points follow in input order (near-coincident points are skipped). */
typedef struct {                 /* VoxelImpactFracturingConfig (fracturing.rs:855-871: the defaults ivx_impact_fracturing_config_default fills in) */
    uint32_t boundary_polar_grid_size, boundary_azimuthal_grid_size;
    float boundary_angular_jitter, boundary_radial_jitter;
    uint64_t max_fragment_count;
    float radial_falloff_power, angular_falloff_power;
    uint32_t radial_grid_size, angular_grid_size;
    uint64_t max_position_rejections_per_sample;
    uint64_t seed;
} ivx_impact_fracturing_config;  /* 56 bytes */
typedef struct {                 /* FracturingProperties, #[repr(C)] Pod (fracturing.rs:61-86) */
    float fracturing_force, shattering_pressure, fragment_scale, min_fragment_extent, max_fragment_extent;
} ivx_fracturing_properties;     /* 20 bytes */
void ivx_impact_fracturing_config_default(ivx_impact_fracturing_config*);
/* Points in the object's normalized space (voxel units). rng_state: in/out state of the frame's generator (first impact: the seed). The
 * stream is fastrand 2.3.0's wyrand restated from its published source: the crate is not vendored, no reference test pins it. */
int ivx_generate_impact_fracture_points(const ivx_impact_fracturing_config*, const ivx_fracturing_properties*, float inverse_voxel_extent,
                                        const float world_to_object_rotation_xyzw[4], const float world_to_object_translation[3], const float object_aabb[6],
                                        const float force_position[3], const float force_unit_direction[3], float force_magnitude, uint64_t* rng_state,
                                        float* boundary_points3, size_t cap_boundary, size_t* n_boundary, float* fracture_points3, size_t cap_fracture,
                                        size_t* n_fracture);
typedef struct ivx_delaunay ivx_delaunay;
#define IVX_NO_TETRAHEDRON 0xFFFFFFFFu
int ivx_delaunay_construct(const float* points3, size_t n_points, ivx_delaunay** out);
void ivx_delaunay_destroy(ivx_delaunay*);
int ivx_delaunay_counts(const ivx_delaunay*, uint32_t counts[2] /* vertices incl. the 4 ad-hoc ones, tetrahedra */);
int ivx_delaunay_download(const ivx_delaunay*, float* vertices3, uint32_t* tet_vertices4, uint32_t* tet_neighbors4 /* across the face opposite corner c */);
/* displace_vertices (fracturing.rs:996-1002: the tetrahedralization built on the points as given is moved into the region object's frame) */
int ivx_delaunay_displace_vertices(ivx_delaunay*, const float offset[3]);
int ivx_delaunay_aabb(const ivx_delaunay*, float aabb[6]);
int ivx_delaunay_boundary_face_planes(const ivx_delaunay*, float* planes4 /* outward unit normal, displacement */, size_t cap, size_t* n_out);
int ivx_voronoi_polyhedron(const ivx_delaunay*, uint32_t vertex, float* vertices3, size_t cap_vertices, float* rays6 /* origin, unit direction */, size_t cap_rays,
                           float* planes4, size_t cap_planes, size_t n_out[3] /* vertices, rays, planes */);
int ivx_voronoi_bounded_aabb(const float* vertices3, size_t n_vertices, const float* rays6, size_t n_rays, const float bounding_aabb[6], float out_aabb[6], int* has);

/* ---- multi-GPU: the whole per-step protocol behind the C ABI ----------------------------------------------------------------------
 * One process per GPU owns one x-slab (an ivx_grid created with its chunk offset) and an RCCL communicator; ivx_slabs_step_enqueue
 * runs sample -> face exchange -> derive + regions + moments -> face exchange (+ component ids) -> remesh -> record all-gather, all
 * stream-ordered on the context's stream (grouped ncclSend / ncclRecv with the two x neighbours, one small ncclAllGather);
 * ivx_slabs_step_collect waits once, finishes the cross-slab union-find of the region equivalences and hands back the global results.
 * A Rust host binds these five entry points and keeps only the launcher (who is rank r, how the 128-byte unique id reaches the ranks).
 *   ivx_comm_unique_id   rank 0 makes the id (ncclGetUniqueId) and distributes it by its own means
 *   ivx_comm_init        ncclCommInitRank on the context's device; librccl is opened at run time (no link-time dependency)
 *   ivx_comm_init_local  an in-process communicator: all `nranks` slabs live in THIS process on one GPU and exchange by device copies —
 *                        the same driver code, used to check the decomposition on a single-GPU box
 * slabs: RCCL communicator -> exactly one slab (this rank's); in-process communicator -> all of them, in rank order. */
typedef struct ivx_comm ivx_comm;
typedef struct ivx_slab ivx_slab;
typedef struct {
    uint32_t region_count;           /* connected regions of the whole grid */
    uint32_t local_region_count;     /* components of this slab */
    uint32_t first_local_component;  /* index of this slab's component 0 in the concatenation over ranks */
    uint32_t reserved;
    double moments[10];              /* of the whole grid, summed in rank order */
    uint32_t occupied[12];           /* global occupied chunk / voxel ranges (layout of ivx_step_result::occupied) */
    ivx_mesh_counts mesh;            /* this slab's mesh */
    uint64_t vertex_offset, index_offset; /* of this slab's mesh in the concatenation over ranks */
    uint64_t total_triangles;
    float stage_ms[IVX_N_TIMED_STAGES];
    float reserved2[2];
} ivx_slab_result;
int ivx_comm_unique_id(void* out128);
int ivx_comm_init(ivx_ctx*, int nranks, int rank, const void* unique_id128, ivx_comm** out);
int ivx_comm_init_local(ivx_ctx*, int nranks, ivx_comm** out);
/* One PROCESS per rank on ONE device (RCCL refuses two ranks on a device): a rank copies its face planes straight into its neighbour's receive
 * buffer (hipIpcGetMemHandle / hipIpcOpenMemHandle), sequence numbers and the record gather go through the POSIX shared-memory block `name`
 * ("/..."; rank 0 creates it). The host waits at every exchange: a transport for running the protocol's driver as separate processes where
 * there is one GPU (tests/test_gpu_slabs_ipc.py), not for speed. ivx_slab_create is collective on such a communicator. */
int ivx_comm_init_ipc(ivx_ctx*, int nranks, int rank, const char* name, ivx_comm** out);
/* transport (0 RCCL, 1 in-process, 2 shared device), rank count and this process's rank as the communicator reports them (RCCL: ncclCommCount,
 * ncclCommUserRank) */
int ivx_comm_info(ivx_comm*, int* transport, int* nranks, int* rank);
/* Diagnostic (no reference counterpart): the in-process transport normally moves nothing (a slab reads its neighbour's send buffer in place).
 * on = 1: it moves its messages as the RCCL transport does — copies on the communicator's own stream behind the packing, the slabs' derive
 * sweep and mesher count split around their arrival (interior chunk planes first, the planes beside the ghost layers after the wait) — so
 * that a one-GPU box exercises the overlapped protocol; on = 2: the copies on the context's stream, sweeps unsplit; 0: back to in place. */
int ivx_comm_set_local_copies(ivx_comm*, int on);
void ivx_comm_destroy(ivx_comm*);
/* Diagnostic (no reference counterpart): runs every RCCL call the protocol makes — ncclGetUniqueId, ncclCommInitRank, a grouped
 * ncclSend / ncclRecv pair, ncclAllGather — on a ONE-rank communicator of this context's device and stream (the send goes to the rank
 * itself) and checks the bytes that come back. A single-GPU box can so prove that the run-time binding to librccl (dlopen, symbols,
 * argument layouts, the library's stream) works before a multi-rank job depends on it. IVX_OK = all of it worked. */
int ivx_comm_selftest(ivx_ctx*);
/* Diagnostic (no reference counterpart): the mesher divides decoded distances (surface_nets.rs:396-404, t = d1 / (d1 - d2)) and edge counts
 * with a short correctly-rounding sequence instead of the general f32 division; this runs both over the whole operand set (every pair of
 * decoded i8 distances of opposite sign, 1 / n for n = 1..256) on the device and returns the number of results that differ: must be 0. */
int ivx_selftest_mesher_division(ivx_ctx*, uint32_t* mismatches);
int ivx_slab_create(ivx_comm*, ivx_grid* slab_grid, int rank, ivx_slab** out);
void ivx_slab_destroy(ivx_slab*);
int ivx_slabs_step_enqueue(ivx_slab** slabs, size_t n);
int ivx_slabs_step_collect(ivx_slab** slabs, size_t n, ivx_slab_result* out /* n results */);
/* global region id of every slab-local component of `rank` (after a collect; `first_slab` = slabs[0] of that call) */
int ivx_slab_region_map(ivx_slab* first_slab, int rank, uint32_t* out, size_t cap, size_t* n_out);

/* ---- a15-a19: rigid bodies + sequential-impulses contact solve (engine/crates/impact_physics) ----- */
/* DynamicRigidBody, #[repr(C)], 152 bytes (src/rigid_body.rs:94-103). Matrices are Matrix3C (column-major),
 * orientation is UnitQuaternionC (x, y, z, w). */
typedef struct {
    float mass;
    float inertia[9], inv_inertia[9]; /* InertiaTensorC about the centre of mass, body frame (src/inertia.rs:41-47) */
    float position[3];
    float orientation[4];
    float momentum[3], angular_momentum[3];
    float total_force[3], total_torque[3];
} ivx_rigid_body;
/* KinematicRigidBody, 56 bytes (src/rigid_body.rs:108-117): AngularVelocityC = unit axis + angular speed */
typedef struct {
    float position[3];
    float orientation[4];
    float velocity[3];
    float angular_axis[3];
    float angular_speed;
} ivx_kinematic_body;
/* One ContactWithID of a collision (src/constraint/contact.rs:23-57) with the two rigid bodies it acts on.
 * body_a/body_b index the dynamic bodies; bit 31 set = index into the kinematic bodies. position =
 * ContactGeometry::position (point on B), normal = surface normal of B, depth = penetration depth; the
 * three response parameters are the already combined ones (src/material.rs:43-52). flags bit 0 marks the
 * first contact of a manifold (one Collision): interlock analysis works per manifold
 * (contact.rs:610-689). 64 bytes. */
typedef struct {
    uint64_t id; /* ContactID */
    uint32_t body_a, body_b;
    float position[3];
    float normal[3];
    float depth;
    float restitution, static_friction, dynamic_friction;
    uint32_t flags, reserved;
} ivx_contact;

/* ---- voxel contact generation (sphere collidables) -----------------------------------------------------------
 * for_each_sphere_voxel_object_contact (impact_voxel/src/collidable.rs:1098-1127) wrapped as generate_sphere_voxel_object_contact_manifold
 * does (collidable.rs:1051-1096): every surface voxel (non-empty, fewer than six neighbours) in the voxel ranges the sphere's box touches
 * is a sphere of radius -sd * extent tested against the collidable (determine_sphere_sphere_contact_geometry, impact_physics
 * collision/collidable/sphere.rs:105-136). `rotation_xyzw` + `translation` = transform_to_object_space (Isometry3: world -> the object's
 * model space), the sphere is in world space; `response` = the combined restitution, static and dynamic friction. Contacts come in the
 * reference's traversal order (chunks i,j,k then voxels i,j,k) with id = ContactID::from_two_u64_and_n_indices(collidable_id_a,
 * collidable_id_b, [i,j,k]) and the first one flagged IVX_CONTACT_MANIFOLD_START: ready for ivx_world_set_contacts. Needs current derived
 * state. *n_out is the number found even when it exceeds `cap` (IVX_ERR_CAPACITY then). */
int ivx_sphere_voxel_object_contacts(ivx_grid*, const float rotation_xyzw[4], const float translation[3], const float sphere_center[3],
                                     float sphere_radius, uint64_t collidable_id_a, uint64_t collidable_id_b, uint32_t body_a, uint32_t body_b,
                                     const float response[3], ivx_contact* out, size_t cap, size_t* n_out);
/* for_each_voxel_object_plane_contact (collidable.rs:1176-1208): the same for a plane collidable (unit normal + displacement, world space);
 * only Corner voxels (at most three neighbours) inside the voxel ranges of the plane's negative halfspace are tested
 * (determine_sphere_plane_contact_geometry, sphere.rs:138-160). The reference hashes the ids as (plane, voxel object) and the contact
 * normal is the plane's: pass the collidable ids in that order and the voxel object's body as body_a. */
int ivx_plane_voxel_object_contacts(ivx_grid*, const float rotation_xyzw[4], const float translation[3], const float plane_unit_normal[3],
                                    float plane_displacement, uint64_t collidable_id_a, uint64_t collidable_id_b, uint32_t body_a, uint32_t body_b,
                                    const float response[3], ivx_contact* out, size_t cap, size_t* n_out);
/* for_each_capsule_voxel_object_contact (collidable.rs:1257-1286): the same for a capsule collidable (segment start + vector + radius, world
 * space); surface voxels inside the capsule's box are tested with determine_capsule_sphere_contact_geometry
 * (impact_physics/src/collision/collidable/capsule.rs:212-270). Ids and bodies as for the sphere (collidable first). */
int ivx_capsule_voxel_object_contacts(ivx_grid*, const float rotation_xyzw[4], const float translation[3], const float segment_start[3],
                                      const float segment_vector[3], float capsule_radius, uint64_t collidable_id_a, uint64_t collidable_id_b,
                                      uint32_t body_a, uint32_t body_b, const float response[3], ivx_contact* out, size_t cap, size_t* n_out);
/* The same three generators for MANY objects in one call — the reference's collision pass visits every voxel object of the scene with the
 * collidables near it (impact_voxel/src/collidable.rs:1051-1286) —, one collidable per object: query i goes with grids[i]. The per-object
 * launches are merged (csrc/many.hpp) and the host waits twice for ALL objects (totals, then contacts) instead of twice per object.
 * out_offsets has n + 1 entries: object i's contacts are out[out_offsets[i] .. out_offsets[i + 1]), the very list the single-object call
 * returns (one manifold each). IVX_ERR_CAPACITY when the contacts of all objects exceed `cap` (out_offsets then holds the sizes needed). */
typedef struct ivx_collidable_query {
    int32_t mode;             /* 0 sphere, 1 plane, 2 capsule */
    float rotation_xyzw[4];   /* transform_to_object_space of the voxel object, as in the single-object calls */
    float translation[3];
    float shape3[3];          /* sphere centre | plane unit normal | capsule segment start (world space) */
    float shape3b[3];         /* capsule segment vector (unused otherwise) */
    float shape1;             /* sphere radius | plane displacement | capsule radius */
    uint32_t body_a, body_b;
    uint64_t collidable_id_a, collidable_id_b;
    float response[3];        /* restitution, static friction, dynamic friction */
    uint32_t reserved;
} ivx_collidable_query;
int ivx_voxel_object_contacts_many(ivx_grid* const* grids, size_t n, const ivx_collidable_query* queries, ivx_contact* out, size_t cap, uint32_t* out_offsets);

/* ---- contacts between two voxel objects (second part) -------------------------------------------------------------
 * VoxelObjectCollisionProbes::recompute_for_all_chunks (collidable.rs:361-392, 451-523, 614-731): per chunk submesh of the current mesh
 * (ivx_remesh), the mesh vertex of the most convex curvature in every block of 8^3 / 4^3 / 2^3 / 1 voxels (block size from the object's smallest
 * occupied extent). The probes stay on the device with the grid and go stale with the mesh. */
int ivx_collision_probes_recompute(ivx_grid*, size_t* n_points);
/* points: 3 floats each (the whole buffer; after a sync it may contain freed ranges), in submesh order then block order after a recompute; chunk_entries:
 * 5 u32 per chunk that has probes (ci, cj, ck, first point, end point) — the reference's chunk_point_ranges, sorted by first point. Either output may be
 * NULL; the counts are always returned. */
/* VoxelObjectCollisionProbes::sync_with_voxel_object_and_mesh (collidable.rs:394-433, 524-612): after ivx_mesh_sync, with the same invalidated chunks —
 * only their probes are picked again; the point buffer keeps its length or grows, a chunk's points go to the best-fitting range freed before or to
 * the end (RangeAllocator); chunks are visited in chunk-linear order (the reference's hash-set order is unpinned). n_points = length of the buffer,
 * freed ranges included. */
int ivx_collision_probes_sync(ivx_grid*, const uint8_t* invalidated_chunks, size_t* n_points);
/* ivx_collision_probes_sync for N objects of one context in merged launches (cf. ivx_mesh_sync_many, whose invalidated sets these are): the select
 * passes of all objects, one wait, the range allocators on the host, the gathers of all objects, one wait. n_points: one length per object. */
int ivx_collision_probes_sync_many(ivx_grid* const* grids, size_t n, const uint8_t* const* invalidated_chunks, size_t* n_points);
int ivx_collision_probes_download(ivx_grid*, float* points, size_t cap_points, uint32_t* chunk_entries, size_t cap_entries, size_t* n_points,
                                  size_t* n_entries);
/* for_each_mutual_voxel_object_contact (collidable.rs:859-1049): the probes of A inside the voxel ranges where the two occupied boxes can overlap
 * (determine_voxel_ranges_encompassing_intersection, object/intersection.rs:706-746) are sampled against B's signed distance field
 * (determine_sdf_value_and_normal_at_point_if_intersecting, collidable.rs:1288-1440), then B's probes against A. rotation (xyzw) + translation =
 * each object's transform_to_object_space (world -> object), center_of_mass_* = derive_center_of_mass() of the object's inertial properties
 * (object space, world units). Every contact hashes [0, i, j, k] of the probing object's voxel with the two collidable ids; normals point from B
 * to A. Both objects need current derived state and current probes, and must belong to the same context. The reference walks the chunks in the
 * iteration order of a hash map (unpinned, see oracle/src/orc_collide.cpp); here they come in submesh order, A's probes first. */
int ivx_mutual_voxel_object_contacts(ivx_grid* a, const float rotation_a[4], const float translation_a[3], const float center_of_mass_a[3], ivx_grid* b,
                                     const float rotation_b[4], const float translation_b[3], const float center_of_mass_b[3],
                                     uint64_t collidable_id_a, uint64_t collidable_id_b, uint32_t body_a, uint32_t body_b, const float response[3],
                                     ivx_contact* out, size_t cap, size_t* n_out);
/* The same for a list of pairs in merged launches with two waits for all (cf. ivx_voxel_object_contacts_many): the pairs the broad phase found
 * this frame. Pair i's manifold is out[out_offsets[i] .. out_offsets[i + 1]) (n + 1 offsets), the list the single-pair call returns. An object
 * may appear in any number of pairs; the probes of every object must be current. */
typedef struct ivx_mutual_query {
    ivx_grid* a;
    ivx_grid* b;
    float rotation_a[4], translation_a[3], center_of_mass_a[3];
    float rotation_b[4], translation_b[3], center_of_mass_b[3];
    uint64_t collidable_id_a, collidable_id_b;
    uint32_t body_a, body_b;
    float response[3];
    uint32_t reserved;
} ivx_mutual_query;
int ivx_mutual_voxel_object_contacts_many(const ivx_mutual_query* queries, size_t n, ivx_contact* out, size_t cap, uint32_t* out_offsets);

#define IVX_KINEMATIC_BODY 0x80000000u
#define IVX_CONTACT_MANIFOLD_START 1u
/* ConstraintSolverConfig (src/constraint/solver.rs:41-57; defaults 8, 0.4, 3, 0.2 at 374-384) */
typedef struct {
    uint32_t n_iterations;
    float old_impulse_weight;
    uint32_t n_positional_correction_iterations;
    float positional_correction_factor;
} ivx_solver_config;
typedef struct {
    uint32_t n_contacts;   /* prepared contacts (after interlock replacement and cache clean-up) */
    uint32_t n_bodies;     /* constrained bodies (touched by >= 1 contact) */
    uint32_t n_levels[2];  /* dependency levels of the velocity and the positional schedule */
    float stage_ms[5];     /* prepare, pre-solve (momenta + velocity sync), solve, write-back + configurations, total */
    uint32_t reserved[3];
} ivx_physics_result;

/* ---- a14: rigid bodies after voxels were removed (impact_voxel/src/interaction.rs:224-602) ----------------------------------------------
 * Host arithmetic, O(1) per object. `moments` are the ten f64 moments of a VoxelObjectInertialPropertyManager about the object's grid origin
 * (ivx_moments.m64 / ivx_region_desc.moments: mass, first moments, moments of inertia, products xy, yz, zx).
 * apply_updated_inertial_properties_to_rigid_body (interaction.rs:405-458; preserve_momentum = 1: ..._preserving_momentum, 460-487): new mass and
 * inertia tensor, position moved by the rotated shift of the local centre of mass, momenta re-synchronised for v + w x shift and the unchanged
 * angular velocity. */
/* VoxelObjectInertialPropertyManager::offset_reference_point_by (object/inertia.rs:257-267): the same moments about the point `offset` (e.g. a fragment's
 * moments from its own grid frame into the parent's: offset = -origin_offset_in_parent * voxel_extent) */
int ivx_offset_reference_point(double moments[10], const float offset[3]);
int ivx_apply_updated_inertial_properties(ivx_rigid_body* body, const double moments[10], const float original_local_center_of_mass[3],
                                          int preserve_momentum, float new_local_center_of_mass[3]);
/* determine_extracted_voxel_object_dynamics (interaction.rs:503-585): `moments` in = the fragment's moments in the PARENT's grid frame (what
 * ivx_split_off_smallest_region reports in `moved`), out = about the fragment's own grid origin (offset_reference_point_by, object/inertia.rs:257-267);
 * fragment_body = DynamicRigidBody::new(mass, inertia, position of its own centre of mass, parent orientation, v + w x shift, w). */
int ivx_extracted_object_dynamics(double moments[10], const uint32_t origin_offset_in_parent[3], float voxel_extent,
                                  const float original_local_center_of_mass[3], const ivx_rigid_body* parent_body, ivx_rigid_body* fragment_body,
                                  float new_local_center_of_mass[3]);
typedef struct {
    ivx_grid* grid; /* the new object; the caller destroys it */
    uint32_t origin_offset_in_parent[3];
    uint32_t reserved;
    ivx_rigid_body body;
    double moments[10]; /* its inertial property manager, about its own grid origin */
    float local_center_of_mass[3];
    float reserved2;
} ivx_extracted_object;
/* handle_voxel_object_after_removing_voxels (interaction.rs:224-403) without the anchor bookkeeping: while the object has disconnected regions the
 * smallest is split off (ivx_split_off_smallest_region), its moments leave `moments` and, if it is a new object, it gets its rigid body from the
 * parent body's state BEFORE this call; finally the parent body takes the remaining inertial properties (momentum re-synchronised unless
 * removed_mass_destroyed is set and nothing was split off). `moments` in = the object's manager after the removal that triggered the call.
 * original_object_empty = the object has fewer than 8 non-empty voxels left (is_effectively_empty, object.rs:803-845); its body is then left as is. */
int ivx_handle_voxel_object_after_removing_voxels(ivx_grid*, const float densities[256], double moments[10], ivx_rigid_body* body,
                                                  const float original_local_center_of_mass[3], int removed_mass_destroyed, ivx_extracted_object* out,
                                                  size_t cap, size_t* n_out, int* original_object_empty, float new_local_center_of_mass[3]);

/* RigidBodyManager + ConstraintManager of one simulation (src/rigid_body.rs:71-78, src/constraint.rs:33-39) */
typedef struct ivx_world ivx_world;
int ivx_world_create(ivx_ctx*, const ivx_solver_config*, ivx_world** out);
void ivx_world_destroy(ivx_world*);
int ivx_world_set_bodies(ivx_world*, const ivx_rigid_body* dynamic, size_t n_dynamic, const ivx_kinematic_body* kinematic, size_t n_kinematic);
int ivx_world_get_bodies(ivx_world*, ivx_rigid_body* dynamic, ivx_kinematic_body* kinematic); /* either may be NULL */
/* The collisions of this step (what CollisionWorld hands to ConstraintManager::prepare_constraints,
 * src/constraint.rs:193-265). Host side: interlock replacement, the ConstraintCache bookkeeping that fixes
 * the solve order and the warm-start source of every contact (solver.rs:386-452), and the dependency
 * schedule that lets the GPU run the reference's sequential sweeps in parallel with identical results. */
int ivx_world_set_contacts(ivx_world*, const ivx_contact*, size_t n, size_t* n_prepared);
/* SphericalJoint constraints (src/constraint/spherical_joint.rs, ConstraintManager::add_spherical_joint src/constraint.rs:183-190): two body
 * references per joint (IVX_KINEMATIC_BODY flag as in ivx_contact). The reference's joint is a placeholder that applies no impulse and no
 * positional correction (spherical_joint.rs:62-88); preparing it only registers its bodies as constrained bodies of the step, whose
 * velocities are written back after the solve — which is all this entry point makes happen. Stays in force until replaced; call it after
 * ivx_world_set_bodies. */
int ivx_world_set_spherical_joints(ivx_world*, const uint32_t* body_pairs, size_t n_joints);
/* perform_physics_step (src/lib.rs:31-110) without force generators / motion drivers: prepare constraints ->
 * advance momenta -> synchronise velocities, warm start, n_iterations sweeps, positional correction,
 * write back -> advance configurations. The stages are also exposed one by one. */
int ivx_world_step(ivx_world*, float dt, ivx_physics_result* out);
/* the same step, only enqueued on the context's stream (no wait, no result): lets the rigid-body step of a frame run
 * behind the voxel step of the same frame with a single wait for both (ivx_voxel_step_collect / ivx_synchronize);
 * the bodies are read back with ivx_world_get_bodies when they are needed */
int ivx_world_step_enqueue(ivx_world*, float dt);
int ivx_world_prepare(ivx_world*);
int ivx_world_advance_momenta(ivx_world*, float dt);
int ivx_world_solve(ivx_world*);
int ivx_world_advance_configurations(ivx_world*, float dt);
/* ContactIDs in solve order and their accumulated (normal, tangent, bitangent) impulses after the last solve */
/* How the solve walks the exact-order schedule. groups = 0 (default): chosen from the schedule — one workgroup with the bodies in LDS when the
 * widest dependency level fits it, else the chain-stationary solve (every chain of contacts keeps one lane of one wave for the whole phase with
 * its prepared contacts and accumulated impulses in registers; only the two bodies' state travels, through version-tagged records; both phases
 * in one launch), else tiles of the level schedule handed to up to 16 workgroups. 1: one workgroup. 2..16: the tile form on that many
 * workgroups. 255: the chain-stationary solve whatever the widths (tests). Same results whichever runs. ivx_world_solver_info: out[0]
 * workgroups used by the last solve, [1..2] levels of the velocity / positional schedule, [3..4] widest level of each, [5] chains (runs of
 * up to 4 contacts of one manifold), [6] contacts, [7] the kernel that ran (0 one workgroup, 1 tiles, 2 chain-stationary). */
int ivx_world_set_solver_groups(ivx_world*, uint32_t groups);
int ivx_world_solver_info(ivx_world*, uint32_t out[8]);
int ivx_world_contact_state(ivx_world*, uint64_t* ids, float* impulses3, size_t cap, size_t* n_out);

#ifdef __cplusplus
}
#endif
#endif
