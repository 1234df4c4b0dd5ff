// Internal definitions shared by the HIP translation units of libimpact_voxel_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/impact_voxel_hip.h"
#include "many.hpp"

// Every stream operation of the library goes through these: while a batch of objects is being recorded (many.hpp) whatever has been
// recorded is issued first, so that the stream sees the operations in program order whether or not they have a merged form.
static inline hipError_t ivx_memset_async(void* p, int v, size_t n, hipStream_t s) {
    (void)ivx_many_break();
    return hipMemsetAsync(p, v, n, s);
}
static inline hipError_t ivx_memcpy_async(void* d, const void* src, size_t n, hipMemcpyKind k, hipStream_t s) {
    (void)ivx_many_break();
    return hipMemcpyAsync(d, src, n, k, s);
}
static inline hipError_t ivx_stream_sync(hipStream_t s) {
    (void)ivx_many_break();
    return hipStreamSynchronize(s);
}
static inline hipError_t ivx_memcpy_sync(void* d, const void* src, size_t n, hipMemcpyKind k) {
    (void)ivx_many_break();
    return hipMemcpy(d, src, n, k);
}
static inline hipError_t ivx_event_record(hipEvent_t e, hipStream_t s) {
    (void)ivx_many_break();
    return hipEventRecord(e, s);
}

#define IVX_CHUNK 16
#define IVX_CHUNK_VOXELS 4096
#define IVX_MAX_FACE_PAIRS 4096

// VoxelFlags (impact_voxel/src/lib.rs:75-101)
#define VF_EMPTY 0x01u
#define VF_X_DN 0x04u
#define VF_Y_DN 0x08u
#define VF_Z_DN 0x10u
#define VF_X_UP 0x20u
#define VF_Y_UP 0x40u
#define VF_Z_UP 0x80u
// VoxelChunkFlags (impact_voxel/src/object.rs:163-188)
#define CF_ONLY_EMPTY 0x40u
#define CF_FULLY_OBSCURED 0x3Fu
#define KIND_VOID 0
#define KIND_UNIFORM 1
#define KIND_NONUNIFORM 2
#define FD_EMPTY 0
#define FD_FULL 1
#define FD_MIXED 2
#define TYPE_DUMMY 255
#define SD_VOID_LIMIT 100

struct ivx_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    int n_cu;  // compute units of the device
    // pinned, device-visible scratch of the many-object calls that bring lists back (ivx_voxel_object_contacts_many: per-object totals, then the
    // contacts themselves, written by the kernels straight into host memory); grown on demand, freed by ivx_shutdown
    void* pinned_scratch;
    void* pinned_scratch_dev;
    size_t pinned_scratch_bytes;
    void* dev_scratch;  // device scratch of the same calls (counts and offsets of every pair of ivx_mutual_voxel_object_contacts_many)
    size_t dev_scratch_bytes;
    hipStream_t aux_stream;  // made on first use (ivx_aux_stream): the sampler's pre-pass one step ahead (ivx_grid_set_sample_ahead)
    void* many_recorder;  // the launch recorder of ivx_many_begin / _flush and its staging ring (many.cpp); made on first use, freed by ivx_shutdown
    int many_error;       // a flush of recorded launches failed on this context (sticky until reported: ivx_many_error)
};

// A device allocation (and / or a pinned host allocation) shared by several grids that came into being together — the fragments of an impact
// (fracturing.rs:1047-1189; docs/voxel_gpu_buffer_pooling.md:44-66: arenas, objects hold a part) —, freed by the last holder (ivx_block_release, ivx_api.hip).
struct ivx_block {
    void* dev;
    void* pinned;
    int refs;
};

struct ivx_grid {
    ivx_ctx* ctx;
    uint32_t cc[3];
    uint32_t n_chunks;
    size_t n_vox;
    float extent;
    uint32_t x_off, gx;  // slab offset in chunks and global chunk count along x
    char* arena;  // the one device allocation the buffers below (all sized by the chunk counts) are carved from
    ivx_block* arena_block;  // non-null: the arena is a part of this shared block (ivx_copy_polyhedra), not an allocation of its own
    ivx_block* host_block;   // non-null: result_host is a part of this shared pinned block
    // planes
    int8_t* sdf;
    uint8_t* type;
    uint8_t* flags;
    uint8_t* llabel;
    ivx_chunk_info* info;
    uint32_t* chunk_bbox;  // per chunk: bit31 valid | min/max i,j,k (4 bits each) of non-empty voxels
    // ghost x-face layers [side]: face planes in (cj,ck)-tiled (j,k) order + chunk info of the layer
    int8_t* ghost_sdf[2];
    uint8_t* ghost_type[2];
    ivx_chunk_info* ghost_info[2];
    uint32_t* ghost_rlabel[2];  // global region node of the neighbour's face voxel (0xFFFFFFFF empty)
    int has_ghost[2];
    // (slab protocol) the event behind the exchange that fills the ghost layers, while nobody has waited for it yet: the next derive sweep /
    // mesher count runs its ghost-free part, makes the stream wait, then runs the rest (a hipEvent_t of the communicator; null: nothing pending)
    void* ghost_event;
    int ghost_split;         // ... in two parts around the wait (else the wait alone, ahead of the sweep)
    void* face_ids_event;    // the same for the neighbour's face ids (the second message of the step's second exchange): waited for by the face-pair pass
    const uint8_t* ghost_ext[2];  // ivx_halo_unpack_enqueue: the ghost layer is read in place from the caller's receive buffer
    // mesh state
    uint32_t* chunk_counts;   // [n_chunks*2]: vertex count, index count
    uint32_t* chunk_offsets;  // [n_chunks*2 + 4]: exclusive vertex/index offsets, then totals v,i,submeshes
    float* positions;
    float* normals;
    uint32_t* indices;
    uint8_t* index_materials;
    uint8_t* vertex_materials;  // 16 B per vertex scratch
    ivx_submesh* submeshes;
    size_t vcap, icap, scap;
    ivx_mesh_counts mesh_counts;
    int mesh_valid;
    // inertia / reductions scratch
    double* partials;  // [n_blocks*16]
    size_t partial_blocks;
    // regions
    uint32_t* rparent;   // [n_chunks*256] global DSF parent per (chunk, local region)
    uint32_t* rcompid;   // [n_chunks*256] component id per node (after resolve)
    uint32_t* rscalar;   // small scalars: [0] region count, [1] error flags
    uint32_t* ccl_scratch;  // [2*n_chunks]: per-chunk root counts and exclusive offsets
    uint32_t* sn_list;      // [n_chunks] uint4 records of the chunks that emit a mesh this remesh, in submesh order (written by k_sn_scan)
    uint32_t* group_sums;   // [(1 + IVX_SN_GROUP_WORDS) * ceil(n_chunks/256) + IVX_SN_TAIL_WORDS]: first-level totals of the two-level scans; then the count of sn_hard and the mesher's list cursors
    uint32_t* sn_walk;      // [5 n_chunks + 2] the order the mesher's main pass walks its list in (sn::SnWalk): records, then list indices
    uint32_t* sn_hard;      // [n_chunks] list entries (submesh order) of the chunks the mesher's main pass hands to its general pass
    uint32_t region_count;
    int regions_valid;
    // list-driven stages: k_chunk_pre settles every chunk whose per-step state follows from the chunk records alone (Void,
    // and Uniform chunks surrounded by Uniform chunks: ~85 % of a solid body's chunks) with one THREAD each and lists the
    // rest; the workgroup-per-chunk kernels then walk that list instead of being launched once per chunk of the grid.
    uint32_t* work_counts;  // [8]: [wc_cur] chunks on the active list; [wc_cur ^ 1] is zero, ready to be the next derive sweep's counter
    uint32_t wc_cur;        // k_chunk_pre appends under the counter it is handed and zeroes the other one: no preset, no memset
    uint32_t scratch_dirty; // IVX_SCRATCH_* groups of small scratch words that a stage has used since they were last preset
    uint32_t* occ_part;     // [12 * ceil(n_chunks / 256)] per-block minima/maxima of the occupied-range reduction (fused step path)
    uint32_t* active_list;  // [n_chunks]
    uint8_t* chunk_class;   // [n_chunks] 1: settled by k_chunk_pre
    uint8_t* kface;         // [n_chunks][4][256] active chunks: the bytes of every (i,j) row on the chunk's two k faces — sdf at k = 0, sdf at
                            // k = 15, type at k = 0, type at k = 15 (k_derive). The mesher's tile load takes its k halo from here: the
                            // planes hold those bytes 16 apart (one cache line per four rows), here consecutive rows are consecutive bytes
    uint16_t* chunk_signs;  // [n_chunks * 256] active chunks: 16-bit "distance negative" mask of every (i,j) row (k_derive), what
                            // the mesher's count pass needs of the 18^3 tile (2 B per row instead of 16 B + two halo bytes)
    uint8_t* chunk_touch;   // [n_chunks] active chunks: bit d set = a voxel pair touches across the +x/+y/+z face (k_derive)
    double* chunk_moments;  // [n_chunks * 10] moments of the NonUniform chunks (fixed summation order whatever the list order)
    uint32_t last_active;   // host: active-list length seen by the last collect (sizes the list-driven grids)
    int planes_compact;  // planes of Void/Uniform chunks may be stale (see ivx_ensure_dense)
    // the sampler wrote `chunk_signs` and `kface` of every chunk it gave planes, all voxels of such a chunk have the type `signs_type`, and
    // nothing has rewritten voxels since (ivx_planes_touched): the derive sweep then reads those 2 + 4 bytes per row instead of the planes
    int signs_current;
    uint8_t signs_type;
    // a box sweep after an edit (ivx_launch_derive_box) has changed chunk kinds since the active list was made: whoever walks the list without
    // a derive sweep of its own rebuilds it first (ivx_ensure_active_list)
    int active_list_stale;
    int regions_labelled_locally;  // (set around the resolve stage of an edit: the box sweep has labelled the chunk-local regions)
    struct ivx_edit_state* edit;   // host state of the edit path (ivx_absorb_*_enqueue / _collect, ivx_mesh_sync_enqueue / _collect)
    // handed to the next k_step_post1 / k_step_gather launch and consumed by it (the edit path's riders on the step's launches)
    uint32_t post1_needs_box[12];         // touched box lo, cc; grown box lo, cc
    const uint32_t* post1_needs_touched;
    uint32_t* post1_needs_out;            // non-null: the next post1 launch hosts the mesh-needs role
    // ... which also delivers its records EARLY (ivx_mesh_sync_enqueue with a null set while the edit is in flight): straight into host-mapped
    // memory, the role's last workgroup ringing a bell behind them — the host places the invalidated meshes while the region stages still run
    int early_needs_on;                   // ivx_grid_set_early_mesh_needs: the edits of this object deliver early (a system-scope fence per workgroup of the role: ~4 us per edit)
    uint32_t* post1_needs_early;          // host-mapped twin of post1_needs_out (null: no early delivery)
    uint32_t* post1_needs_counter;        // workgroups of the role that are through (device word, zero at the start)
    uint32_t* post1_needs_bell;           // host-mapped: post1_needs_seq once all are
    uint32_t post1_needs_seq;
    const uint32_t* gather_copy_src;
    uint32_t* gather_copy_dst;            // (device address of host-mapped memory)
    uint32_t gather_copy_words;
    uint64_t gather_flush_id;      // the recorder's flush count when the gather was recorded (many.hpp)
    int gather_launched;           // the step's gather is on the stream (or recorded): ivx_voxel_step_collect only waits
    int sn_tail_zero;  // the mesher's hand-off counter, list cursors and census word (IVX_SN_TAIL_WORDS) are known to be zero (ivx_launch_sn_emit_list)
    int needs_current;             // edit->needs holds what the last edit's invalidated chunks' meshes need (no voxel has changed since)
    uint32_t stage_timing_off;  // timed slots WITHOUT event records (ivx_grid_set_stage_timing; zero-initialised: every slot is timed)
    float* dens_call;       // [256] a density table handed to one call (ivx_inertia, ivx_regions_describe) that is not the resident one
    float* dens_dev;        // [256] voxel type densities
    float dens_host[256];   // what dens_dev holds (entry points that are handed the same table again skip the upload)
    void* dev_scratch;      // grown on demand (node programs, dense label export, region statistics)
    size_t dev_scratch_bytes;
    // resident SDF program (ivx_grid_set_sdf_program)
    ivx_sdf_processed_node* prog_nodes;
    uint32_t prog_n, prog_cap, prog_stack;
    uint32_t prog_shape[3];
    float prog_center[3];
    uint8_t prog_type;
    // lengths of the sampler's three evaluation lists under the resident program, as the last collected step with a derive sweep reported them
    // (`eval_len_valid`; a function of the program and the grid alone): a class whose list is known to be empty is not launched
    uint32_t eval_len[3];
    int eval_len_valid, eval_len_pending;
    int has_dens;
    double* moments_dev;  // [10]
    uint32_t* samp_len;   // [n_chunks] length of the chunk's compact SDF program (sampler pre-pass)
    void* samp_ops;       // [n_chunks * 128] uint2 ops
    // Sample-ahead (ivx_grid_set_sample_ahead, sdf_sample.hip): the pre-pass of the NEXT sample stage under the resident program runs on the
    // context's second stream behind this step's evaluator, into a second set of the sampler's buffers; the records it would write for constant
    // chunks wait in `info_shadow` until the next evaluator launch commits them.
    int ahead_on, ahead_pending, ahead_wanted, ahead_unordered_ok, shadow_sel, alt_eval_dirty, ahead_events_ready;
    uint32_t* samp_len_alt;
    void* samp_ops_alt;
    ivx_chunk_info* info_shadow;
    hipEvent_t ahead_go, ahead_done;
    uint32_t* samp_super;  // [super-blocks * ceil(nodes / 32)] "node certainly outside every chunk of the super-block" bits
    size_t samp_super_words;
    hipEvent_t ev[2 * IVX_N_TIMED_STAGES];  // start/stop per timed stage
    int ev_ready;
    uint32_t* result_host;      // 64 words of host-mapped pinned memory the gather kernel writes the step's small results to
    uint32_t* result_host_dev;  // its device-side address
    uint32_t result_seq;        // sequence number of the last gather launch (word 63 of the block = the completion doorbell)
    hipEvent_t* ev_start_ref[IVX_N_TIMED_STAGES];  // the event a stage's duration starts from (the previous stage's stop when adjacent)
    uint32_t pending_stages;  // stages enqueued since the last collect
    uint32_t timed_mask;      // timed stages whose events were recorded since the last collect
    uint32_t* pairs_dev;      // [4 + 128 + 2 * IVX_MAX_FACE_PAIRS]: count, seen table, (own, neighbour) component pairs across the upper x face
    unsigned long long* record_head_copy;  // (in-process slab transport) where the record role also puts the record's first `record_head_words` words:
    uint32_t record_head_words;            // this slab's place in the gathered block — the gather of the heads is then no copy at all
    int results_in_block;     // the step's small results are in the host-mapped block already (written by the slab record role): collect launches no gather
    int pairs_enqueued;
    int pairs_zeroed;         // the label pass of ivx_halo_pack_both_enqueue cleared count + seen table for the face-pair pass that follows
    // scratch groups a caller that knows its next call (the slab protocol) has preset one call ahead, by a kernel that is launched anyway
    uint32_t preset_ahead, preset_fresh;
    // host pinned scratch
    void* host_scratch;
    size_t host_scratch_bytes;
    // collision probes (ivx_collision_probes_recompute): points in submesh order, the chunk of every point (ci | cj << 10 | ck << 20),
    // one entry (ci, cj, ck, first, end) per submesh
    float* probe_points;
    uint32_t* probe_chunk;
    uint32_t* probe_entries;
    size_t probe_point_cap, probe_entry_cap;
    uint32_t n_probe_points, n_probe_sub;
    uint64_t mesh_serial, probes_serial;  // probes are current while they were picked from the current mesh
    // mesh arrays that are parts of a shared block instead of allocations of their own (the first mesh of fragments stepped together,
    // ivx_voxel_step_many; the developer experiment IVX_MESH_ARENA): bit 0 positions + normals + vertex scratch, bit 1 indices + index materials,
    // bit 2 submeshes. A group that grows moves out into allocations of its own; the block goes with its last part (mesh_group_free).
    ivx_block* mesh_block;
    uint32_t mesh_pooled;
    int defer_mesh_growth;  // ivx_voxel_step_collect leaves buffers that are too small to the caller (ivx_voxel_step_many grows all objects' at once)
    int mesh_growth_pending;
    uint64_t mesh_generation;  // bumped whenever a mesh buffer is reallocated: handles exported earlier (ivx_mesh_export) are stale
    // the object's occupied ranges as the reference keeps them (object.rs:1149-1280): refreshed by an explicit update, by split / clip, and by an
    // edit only when it removed a chunk (intersection.rs:255-257, 384-386, 520-522) — in between they may be wider than the voxels need, and the
    // edit, contact and probe entry points must see exactly those ranges
    uint32_t occ_ref[12];
    int occ_ref_valid;
    int bbox_valid;  // the per-chunk boxes the occupied ranges are reduced from exist (written by the derive sweep; the sampler uses the buffer as scratch)
    int mesh_built;  // the mesh buffers hold a mesh of this grid, current (mesh_valid) or made stale by an edit — what ivx_mesh_sync patches
    struct ivx_submesh_manager* submesh_manager;
    struct ivx_probe_manager* probe_manager;  // host mirror of the probes' chunk -> point range map and range allocator  // host mirror of the ChunkSubmeshManager, built by the first ivx_mesh_sync after a full remesh
};

// host-side description of one pass of the mutual contact generation (see collide.hip)
struct ivx_mutual_pass {
    float center_s[3];        // sampled object's centre of mass, normalized
    float q_s[4], t_s[3];     // world -> sampled object
    float q_p[4], t_p[3];     // world -> probing object
    float box_lo[3], box_hi[3];
    uint32_t clo[3], chi[3];
    int negate;
    uint64_t id_ab;
    uint32_t body_a, body_b;
    float response[3];
};

void ivx_set_error(const char* fmt, ...);

#define IVX_HIP_CHECK(expr)                                                                          \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            ivx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return IVX_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)

#define IVX_REQUIRE(cond, code, ...)     \
    do {                                 \
        if (!(cond)) {                   \
            ivx_set_error(__VA_ARGS__);  \
            return code;                 \
        }                                \
    } while (0)

// XCD-aware bijective block -> work-item remap: blocks b and b+8 share an XCD (its L2), so give each
// XCD a contiguous range of chunks (neighbouring chunks then share an L2).
__device__ __forceinline__ uint32_t ivx_xcd_remap(uint32_t b, uint32_t n) {
    uint32_t q = n >> 3, r = n & 7u, x = b & 7u, p = b >> 3;
    uint32_t base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + p;
}

// Compact planes: the four voxel planes of a chunk hold data only while the chunk is NonUniform. A Void or Uniform chunk is
// one voxel, as in the reference's store (VoxelChunk::{Void, Uniform}, object.rs:96-119): maximally outside with the dummy
// type, or maximally inside with `uniform_type`; every neighbour present (flags 0xFC) and one region (label 0) when Uniform.
// Kernels on the step path synthesise those bytes from the 8-byte chunk record; `ivx_ensure_dense` writes them out
// for callers that want whole planes.
__device__ __forceinline__ uint32_t ivx_uniform_sdf(uint32_t kind) { return kind == KIND_UNIFORM ? 0x80u : 0x7Fu; }
__device__ __forceinline__ uint32_t ivx_uniform_type(const ivx_chunk_info& ci) { return ci.kind == KIND_UNIFORM ? (uint32_t)ci.uniform_type : 0xFFu; }
__device__ __forceinline__ uint32_t ivx_uniform_flags(uint32_t kind) { return kind == KIND_UNIFORM ? 0xFCu : (uint32_t)VF_EMPTY; }
__device__ __forceinline__ uint32_t ivx_uniform_label(uint32_t kind) { return kind == KIND_UNIFORM ? 0u : 0xFFu; }

// Sums on the VALU's DPP path. LDS atomics of many lanes on one address cost ~100 cycles per lane on this part (measured:
// the 90 such lane-operations k_derive's first wave used to issue took 5 us, half the life of its workgroup), so counts
// are reduced in registers and only one lane per wave touches LDS.
//   ivx_row16_sum: inclusive prefix sum inside every aligned group of 16 lanes (a DPP row); lane 15 of a row = row total.
//   ivx_wave_sum:  total of the 64 lanes, in every lane.
__device__ __forceinline__ uint32_t ivx_row16_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1, lanes shifted in read 0
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
    return v;
}
__device__ __forceinline__ uint32_t ivx_wave_sum(uint32_t v) {
    v = ivx_row16_sum(v);
    return ((uint32_t)__builtin_amdgcn_readlane((int)v, 15) + (uint32_t)__builtin_amdgcn_readlane((int)v, 31)) +
           ((uint32_t)__builtin_amdgcn_readlane((int)v, 47) + (uint32_t)__builtin_amdgcn_readlane((int)v, 63));
}

// Workgroup timeline probes for the list-driven kernels: IVX_T(g, entry, slot) stores the 100 MHz wall clock of wave 0.
// Compiled in only by `make TRACE=1`; this is how the cost of same-address LDS atomics and of the dependent load chains
// in these kernels was found (rocprofv3 counters alone did not show it).
#ifdef IVX_WG_TRACE
#define IVX_T(g, entry, slot)                                                        \
    do {                                                                             \
        if (threadIdx.x == 0) (g).trace[(size_t)(entry) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define IVX_T(g, entry, slot) \
    do {                      \
    } while (0)
#endif

struct GridView {
    uint32_t cx, cy, cz;
    const int8_t* sdf;
    const uint8_t* type;
    const uint16_t* signs;
    const uint8_t* kface;
#ifdef IVX_WG_TRACE
    unsigned long long* trace;  // developer build (make TRACE=1): 8 timestamps per list entry, see tools/wg_trace.py
#endif
    const int8_t* ghost_sdf[2];
    const uint8_t* ghost_type[2];
    const ivx_chunk_info* info;
    const ivx_chunk_info* ghost_info[2];
};

// A slab's step in two parts around the arrival of its ghost layers (slab_comm.cpp: the exchange runs on a stream of its own): `x_part` of a
// list-driven launch. 0: every listed chunk; IVX_XPART_INTERIOR: the chunks whose x neighbours are all the slab's own (nothing of a ghost
// layer is read); IVX_XPART_FACES: the chunk planes that adjoin a ghost layer.
#define IVX_XPART_ALL 0u
#define IVX_XPART_INTERIOR 1u
#define IVX_XPART_FACES 2u
#ifdef __HIPCC__
__device__ __forceinline__ bool ivx_xpart_skip(const GridView& g, uint32_t x_part, uint32_t ci) {
    if (x_part == IVX_XPART_ALL) return false;
    const bool face = (ci == 0u && g.ghost_sdf[0] != nullptr) || (ci + 1u == g.cx && g.ghost_sdf[1] != nullptr);
    return face != (x_part == IVX_XPART_FACES);
}
#endif

// Entry of the active list: chunk index in the low 24 bits (ivx_grid_create caps the chunk count at 2^24). k_derive adds what
// the later stages would otherwise fetch from the chunk record first (one dependent memory round trip less per workgroup):
// kind, generated kind, "exposed" (NonUniform and not fully obscured: the chunk has a mesh).
#define IVX_LIST_CHUNK(w) ((w) & 0xFFFFFFu)
#define IVX_LIST_KIND(w) (((w) >> 24) & 3u)
#define IVX_LIST_GEN(w) (((w) >> 26) & 3u)
#define IVX_LIST_EXPOSED(w) (((w) >> 28) & 1u)

// Grid of a list-driven launch: about one workgroup per listed chunk, sized from the previous step's list (a longer list is
// covered by the grid-stride walk). Measured on MI355X at 512^3 (5.7k entries): one entry per workgroup is ~10 % faster than
// a resident set of 8 workgroups per CU walking three entries each — the per-entry chain of dependent loads does not
// overlap inside a workgroup. Never fewer workgroups than one thread per chunk of the grid, for the kernels that settle
// per-chunk words in a prologue.
static inline uint32_t ivx_list_grid(const ivx_grid* g) {
    uint32_t n = g->last_active ? g->last_active + g->last_active / 8u + 64u : (uint32_t)g->ctx->n_cu * 32u;
    const uint32_t lo = (g->n_chunks + 255u) / 256u;
    if (n > g->n_chunks) n = g->n_chunks;
    return n < lo ? lo : n;
}

// chunk planes of a slab that adjoin no ghost layer
static inline bool ivx_has_interior_planes(const ivx_grid* g) { return (int)g->cc[0] - (g->has_ghost[0] ? 1 : 0) - (g->has_ghost[1] ? 1 : 0) > 0; }
static inline uint32_t* ivx_wc(const ivx_grid* g) { return g->work_counts + g->wc_cur; }
// to be called by everything that writes voxel planes other than the sampler (uploads, edits, split / clip / repack, raw plane pointers handed out)
static inline void ivx_planes_touched(ivx_grid* g) {
    g->signs_current = 0;
    g->needs_current = 0;
}
// groups of small scratch words that must hold their preset value when a stage starts
#define IVX_SCRATCH_EVAL_ROLL 8u  // role_preset: copy the sampler's list counters to their statistics words and zero them
#define IVX_SCRATCH_REGIONS 1u  // rscalar[0..16): region count, error flags, multi-region chunk count
#define IVX_SCRATCH_SN 2u       // Surface-Nets group totals
// words behind the Surface-Nets group totals: the general pass's counter, then the main pass's eight list cursors at a stride of 32 words
#define IVX_SN_TAIL_WORDS 288u
// words per group of 256 chunks in the mesher's first-level scan block: vertices, indices, submeshes — and padding up to a cache line of its
// own: with the groups' triples side by side the count role's atomics of neighbouring groups met on one line (k_step_post1 22.9 -> 17.8 us on
// the headline body, 66 -> 56 us all-surface)
#define IVX_SN_GROUP_WORDS 32u
#define IVX_SCRATCH_EVAL 4u     // lengths of the sampler's evaluation lists

static inline GridView ivx_view(const ivx_grid* g) {
    GridView v;
    v.cx = g->cc[0];
    v.cy = g->cc[1];
    v.cz = g->cc[2];
    v.sdf = g->sdf;
    v.type = g->type;
    v.signs = g->chunk_signs;
    v.kface = g->kface;
#ifdef IVX_WG_TRACE
    v.trace = reinterpret_cast<unsigned long long*>(g->chunk_moments);  // (the inertia stage is not run while tracing)
#endif
    v.info = g->info;
    for (int s = 0; s < 2; ++s) {
        if (g->has_ghost[s] && g->ghost_ext[s]) {  // message layout: sdf plane | type plane | chunk records
            const size_t cols = (size_t)g->cc[1] * g->cc[2];
            v.ghost_sdf[s] = reinterpret_cast<const int8_t*>(g->ghost_ext[s]);
            v.ghost_type[s] = g->ghost_ext[s] + cols * 256;
            v.ghost_info[s] = reinterpret_cast<const ivx_chunk_info*>(g->ghost_ext[s] + cols * 512);
        } else {
            v.ghost_sdf[s] = g->has_ghost[s] ? g->ghost_sdf[s] : nullptr;
            v.ghost_type[s] = g->has_ghost[s] ? g->ghost_type[s] : nullptr;
            v.ghost_info[s] = g->has_ghost[s] ? g->ghost_info[s] : nullptr;
        }
    }
    return v;
}

// kernels (one launcher per stage; all asynchronous on ctx->stream)
int ivx_launch_classify(ivx_grid* g);
int ivx_launch_sdf_sample(ivx_grid* g, const ivx_sdf_processed_node* d_nodes, uint32_t n_nodes, uint32_t stack_size,
                          const uint32_t shape[3], const float shifted_center[3], uint8_t voxel_type, uint32_t preset_groups = 0,
                          bool resident_program = false);
// parts of the fused sweep: k_derive can label the chunk-local regions and compute the chunk moments of the chunks it visits
#define IVX_PART_REGIONS 1u
#define IVX_PART_MOMENTS 2u
int ivx_launch_derive(ivx_grid* g, uint32_t parts, uint32_t preset_groups = 0);
int ivx_launch_derive_box(ivx_grid* g, uint32_t parts, const uint32_t lo[3], const uint32_t cc[3], uint32_t* d_out_list);
int ivx_ensure_active_list(ivx_grid* g);
int ivx_ensure_dense(ivx_grid* g);
int ivx_launch_step_preset(ivx_grid* g, uint32_t groups);
// scratch groups of the caller's NEXT ivx_voxel_step_enqueue, to be preset by the first kernel of the one before it (slab_comm.cpp)
static inline void ivx_step_preset_ahead(ivx_grid* g, uint32_t groups) { g->preset_ahead |= groups; }
int ivx_voxel_step_enqueue_part(ivx_grid* g, uint32_t stages, uint32_t part);  // (1: the sample and derive sweeps of the call, 2: the rest)
// both faces of a slab into the two message buffers: `what` bit 0 = face planes + chunk records, bit 1 = the face voxels' component ids
int ivx_launch_halo_pack_parts(ivx_grid* g, void* buf_lo, void* buf_hi, uint32_t what);
int ivx_launch_step_post1(ivx_grid* g, uint32_t stages);
int ivx_launch_step_post2(ivx_grid* g, uint32_t stages, const uint16_t* face_pair_ids = nullptr);
int ivx_launch_step_emit(ivx_grid* g, uint32_t stages, bool general_in_assign = false, void* slab_record = nullptr, bool record_has_pairs = false);
int ivx_launch_step_assign(ivx_grid* g, bool with_mesher_general = false, bool with_ccl = true);
int ivx_launch_step_gather(ivx_grid* g);
bool ivx_step_assign_fits(const ivx_grid* g);
int ivx_sampler_buffers(ivx_grid* g);
int ivx_sampler_ahead_cancel(ivx_grid* g);  // sample-ahead (sdf_sample.hip): wait for a pre-pass that runs ahead and drop it
void ivx_sampler_ahead_free(ivx_grid* g);
int ivx_sampler_launch_ahead(ivx_grid* g, bool behind_stream);
void ivx_sdf_annotate_host(ivx_sdf_processed_node* nodes, size_t n);  // sdf_compile.cpp: reserved[] fields for the pre-pass
int ivx_launch_occupied(ivx_grid* g, uint32_t* d_raw);
void ivx_occupied_from_raw(const ivx_grid* g, const uint32_t raw[12], uint32_t out[12]);
int ivx_launch_sn_count(ivx_grid* g);
int ivx_launch_list_mesh_needs(ivx_grid* g, uint32_t n, const uint32_t* d_list, uint32_t* d_out);
int ivx_launch_box_mesh_needs(ivx_grid* g, const uint32_t t_lo[3], const uint32_t t_cc[3], const uint32_t b_lo[3], const uint32_t b_cc[3], const uint32_t* d_touched,
                              uint32_t* d_out);
int ivx_launch_sn_scan(ivx_grid* g);
int ivx_launch_sn_emit(ivx_grid* g);
int ivx_launch_sn_emit_general(ivx_grid* g);
uint32_t* ivx_sn_hard_count(ivx_grid* g);
int ivx_launch_inertia(ivx_grid* g, const float* d_dens, double* d_out10, int fused);
int ivx_launch_ccl_local(ivx_grid* g, int fused);
int ivx_launch_ccl_local_only(ivx_grid* g);  // level 1 over the active list without the exact numbering of multi-region chunks
int ivx_launch_inertia_dense(ivx_grid* g);   // chunk moments of the listed NonUniform chunks into their slots
int ivx_launch_ccl_merge(ivx_grid* g);
int ivx_launch_ccl_resolve(ivx_grid* g);
int ivx_launch_ccl_dense_labels(ivx_grid* g, uint32_t* d_labels);
int ivx_launch_halo_pack(ivx_grid* g, int side, void* buf);
int ivx_launch_halo_pack_both(ivx_grid* g, void* buf_lo, void* buf_hi, int with_face_labels);
int ivx_launch_face_ids(ivx_grid* g, int side, uint16_t* d_out);
int ivx_launch_face_pairs(ivx_grid* g, int side, const uint16_t* d_nbr, uint32_t* d_count, void* d_pairs, uint32_t cap, uint32_t* d_seen);
int ivx_launch_step_record(ivx_grid* g, const uint32_t* d_pair_count, const void* d_pairs, uint32_t max_pairs, void* d_record, bool with_results = false);
int ivx_launch_split_move(ivx_grid* parent, ivx_grid* child, const uint32_t lo[3], const uint32_t cc[3], uint32_t target);
int ivx_launch_clip(ivx_grid* parent, ivx_grid* child, const uint32_t lo[3], const uint32_t cc[3], const float* planes4, uint32_t n_planes, int extract);
int ivx_launch_split_repack(ivx_grid* src, ivx_grid* dst, const uint32_t off[3]);
void ivx_submesh_manager_free(struct ivx_submesh_manager* m);
void ivx_probe_manager_free(struct ivx_probe_manager* m);
int ivx_launch_sn_emit_list(ivx_grid* g, uint32_t n_records, const uint32_t* d_count, const void* d_records, const uint32_t* d_slots, uint32_t n_patch = 0,
                            const void* d_patch_entries = nullptr, const uint32_t* d_patch_slots = nullptr);
int ivx_launch_probe_select(ivx_grid* g, uint32_t n_sub, uint32_t log2_bs, uint32_t* d_corner_list, uint32_t* d_sel, uint32_t* d_counts, uint32_t* d_offsets,
                            uint32_t* d_err, const uint32_t* d_slots, uint32_t* d_counts_host = nullptr);
int ivx_launch_probe_gather(ivx_grid* g, uint32_t n_sub, uint32_t log2_bs, const uint32_t* d_sel, const uint32_t* d_counts, const uint32_t* d_offsets,
                            uint32_t* d_entries, const uint32_t* d_slots);
int ivx_launch_mutual_pass(ivx_grid* prober, ivx_grid* sampled, const ivx_mutual_pass* h, uint32_t* d_counts, const uint32_t* d_offsets, ivx_contact* d_out,
                           uint32_t cap, int emit);
int ivx_launch_scan_counts(ivx_ctx* ctx, uint32_t n, const uint32_t* d_counts, uint32_t* d_offsets, uint32_t* d_total_out = nullptr, const void* owner = nullptr);
int ivx_launch_sdf_snapshot(ivx_grid* g, const int32_t lo[3], const int32_t hi[3], int8_t* d_out);
int ivx_launch_absorb_mutual(ivx_grid* g, int from_snapshot, const uint32_t lo[3], const uint32_t cc[3], const int32_t vlo[3], const int32_t vhi[3],
                             ivx_grid* other, const int8_t* d_snapshot, const int32_t s_lo[3], const int32_t s_hi[3], const float q_ba[4],
                             const float t_ba[3], float smoothness, const float* d_dens, double* d_removed10, uint32_t* d_by_type, uint32_t* d_counters,
                             uint32_t* d_touched);
int ivx_launch_absorb(ivx_grid* g, int capsule, const uint32_t lo[3], const uint32_t cc[3], const int32_t vlo[3], const int32_t vhi[3], const float c[3],
                      const float seg[3], float influence_radius, float shape_radius, const float* d_dens, double* d_removed10, uint32_t* d_by_type,
                      uint32_t* d_counters, uint32_t* d_touched, uint32_t* d_zero16 = nullptr);
int ivx_launch_sphere_contacts(ivx_grid* g, const uint32_t lo[3], const uint32_t cc[3], const int32_t vlo[3], const int32_t vhi[3],
                               const float rotation_xyzw[4], const float translation[3], const float center[3], const float seg_vec[3], float radius,
                               uint64_t id_a, uint64_t id_b, uint32_t body_a, uint32_t body_b, const float response[3], uint32_t* d_counts,
                               uint32_t* d_offsets, uint32_t* d_total, ivx_contact* d_out, uint32_t cap, int emit, int mode);
int ivx_launch_region_stats(ivx_grid* g, const float* d_dens, void* d_buf, uint32_t n);
static inline size_t ivx_region_stats_bytes(uint32_t n) { return (size_t)n * (8 + 80 + 12 + 12 + 4 + 4 + 4); }
