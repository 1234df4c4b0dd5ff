// The fused launches of a voxel step (ivx_voxel_step_enqueue): after the derive sweep a step is a dozen table-sized passes over the
// chunk records (32 768 at 512^3) — occupied ranges, the cross-chunk region merge, flatten, assign, the mesher's count and scan,
// the sum of the chunk moments, the step's small results — plus the mesher's emit pass. Launched one by one, each of the small ones
// costs a launch boundary (~4 us as a kernel, ~1.5 us of gap) for a few hundred KB of work: a third of the 512^3 step. Passes
// that do not depend on each other are therefore hosted as ROLES of one launch: a role owns a range of the launch's block
// indices and runs exactly the code of its stand-alone kernel (ccl_roles.hpp, sn_roles.hpp, table_roles.hpp).
//
//   k_step_post1  after k_derive:   mesher count | region merge by chunk columns | occupied (per-block slots) | moment partial sums
//   k_step_post2  after post1:      exact local numbering -> region merge of multi-region chunks | mesher scan | moments final | occupied final
//                                   (+ the step's results when the call has no region stage)
//   k_step_emit   after post2:      region forest flatten | mesher emit
//   k_step_assign after emit:       component ids (+ the step's results into the host-mapped block)
//
// Dependencies inside a launch: none (roles only read what earlier launches wrote) — but for one, in k_step_post2 (see step_post2_body). The scratch words the stages start from are
// preset by the step's first kernel (k_sdf_super or k_chunk_pre), not by a launch of their own.
#include "ccl_roles.hpp"
#include "sn_roles.hpp"
#include "table_roles.hpp"
#include "many.hpp"

namespace {
using namespace ivx_roles;

struct StepArgs {
    GridView g;
    float extent;
    uint32_t x_off;
    uint32_t n_chunks;
    // block ranges of the roles (0 = the role is not part of this launch)
    uint32_t nb[6];
    // regions
    const uint8_t* flags;
    uint8_t* labels;
    ivx_chunk_info* info;
    uint32_t* rparent;
    uint32_t* rcompid;
    uint32_t* rscalar;
    uint32_t* multi_list;     // = root_counts of the resolve pass afterwards
    uint32_t* ccl_group_sums;
    const uint8_t* touch;
    // mesher
    uint32_t* counts;
    uint32_t* sn_group_sums;
    uint32_t* offsets;
    uint32_t* ranks;
    uint4* emit_items;
    const uint32_t* work_count;
    const uint32_t* active_list;
    float* positions;
    float* normals;
    uint32_t* indices;
    unsigned long long* imats;
    uint4* vmats;
    uint32_t* hard_count;  // chunks the mesher's main pass hands to the general pass: counter, list
    uint32_t* hard_list;
    sn::SnWalk walk;       // the order the main pass takes its list in (role_sn_scan writes it, role_sn_emit reads it)
    ivx_submesh* submeshes;
    uint32_t vcap, icap, scap;
    // occupied / moments
    const uint32_t* bbox;
    uint32_t* occ_part;
    uint32_t n_occ_slots;
    const float* dens;
    const double* chunk_moments;
    double* partials;
    uint32_t n_partials;
    double* moments_out;
    // results
    uint32_t* host_block;     // null: no gather in this launch
    const uint32_t* eval_count;
    uint32_t count_run;       // k_step_post1: list entries per wave and turn of the mesher's count role
    uint32_t x_part;          // k_step_post1, count role: IVX_XPART_* (the slab protocol's split around the arrival of the ghost layers)
    uint32_t seq;             // k_step_gather: the step's sequence number, written last (the host's completion doorbell)
    // slab protocol, remesh phase (ivx_slab_remesh_enqueue): the face-pair pass as a role of k_step_post2, the slab's record (and the step's
    // results into the host-mapped block) as a role of k_step_emit
    uint32_t fp_side, fp_cap;
    const uint16_t* fp_nbr;
    uint32_t* fp_count;  // (null for a slab without an upper neighbour: the record then lists no pairs)
    uint2* fp_pairs;
    uint32_t* fp_seen;
    unsigned long long* record;
    uint32_t record_max_pairs;
    unsigned long long* record_head;  // (optional) a second place for the record's first record_head_words words
    uint32_t record_head_words;
    // edit path: what the meshes of the chunks an edit invalidates need, as a role of k_step_post1 (ivx_grid::post1_needs); a block of small
    // results copied to host-mapped memory by k_step_gather ahead of the doorbell (ivx_grid::gather_copy_*)
    sn::BoxNeeds needs_box;
    const uint32_t* needs_touched;
    uint32_t* needs_out;
    uint32_t* needs_early;    // (optional) host-mapped twin of needs_out, a counter of finished workgroups, the bell behind them and its value
    uint32_t* needs_counter;
    uint32_t* needs_bell;
    uint32_t needs_seq, pad_needs_;
    const uint32_t* copy_src;
    uint32_t* copy_dst;
    uint32_t copy_words;
};

__device__ __forceinline__ sn::SnParams sn_params(const StepArgs& a) {
    sn::SnParams p;
    p.g = a.g;
    p.extent = a.extent;
    p.x_off = a.x_off;
    return p;
}

// roles: 0 mesher count (list-driven), 1 region merge by columns, 3 occupied slots, 4 moment partial sums,
// 5 (edit path) mesh needs of the chunks the edit invalidates
// (Registers: the count role is a latency-bound gather with little state and was given all eight workgroups a CU can hold — 64 VGPRs — in
// round 3. Held to 64 the launch spills 17 registers per lane, and a spilled dword is a store to memory on this part: 24 MB of writes per
// headline launch, 44 MB on the all-surface grid, for a kernel whose own output is a few hundred KB. At six waves per SIMD (80 registers,
// nothing spilled) it writes 0.8 / 9 MB and takes 23.4 instead of 25.2 us, 63.5 instead of 75.5 (tools/post1_waves.sh, profiles/round5).
// The exact numbering of multi-region chunks was a role of this launch until round 4: held to 64 registers it spilled 160 words and a chunk
// took 15-90 us — it now leads k_step_post2, see there.)
__device__ __forceinline__ void step_post1_body(const StepArgs& a, uint32_t b, uint32_t) {
    __shared__ uint32_t sh_words[4 * sn::NROWS + 16];  // (tile sign rows: a wave each in the count role, one set in the needs role)
    struct { uint32_t* par; } sh{sh_words};
    // (the region merge's blocks come first: it is the role with the longest chain of dependent loads and atomics — 19 us by itself on the
    // headline body against the count's 13 — and workgroups start in index order: behind the count role's blocks, which fill every slot
    // of the chip, its waves used to start when the first of those were through)
    if (b < a.nb[1]) {
        role_ccl_merge_columns(b, a.nb[1], a.g, a.touch, a.rparent);
        return;
    }
    b -= a.nb[1];
    if (b < a.nb[0]) {
        // (the class counts of the main pass's walk order, which the scan role of the launch behind this one adds to: zero from here — words of their
        // own, not of the mesher's tail block, which a re-emit after the buffers grew clears while the walk order still stands)
        if (b == 0u && threadIdx.x < 2u && a.walk.items) a.walk.count[threadIdx.x] = 0u;
        sn::role_sn_count_waves(b, a.nb[0], sn_params(a), a.counts, a.sn_group_sums, a.work_count, a.active_list, sh.par, a.count_run, a.x_part);
        return;
    }
    b -= a.nb[0];
    if (b < a.nb[3]) {
        role_occupied_partial(b, a.g.cx, a.g.cy, a.g.cz, a.bbox, a.occ_part);
        return;
    }
    b -= a.nb[3];
    if (b < a.nb[4]) {
        role_inertia_sum(b, a.nb[4], a.g, a.x_off, a.dens, a.chunk_moments, a.partials);
        return;
    }
    b -= a.nb[4];
    if (b < a.nb[5]) {
        sn::role_box_mesh_needs(b, sn_params(a), a.needs_box, a.needs_touched, nullptr, a.needs_out, sh.par);
        // early delivery (ivx_mesh_sync_enqueue while the edit is in flight): the chunk's record a second time, straight into host-mapped
        // memory; the workgroup that finds all others through rings the bell. (Thread 0 wrote the record: it reads its own stores.)
        if (a.needs_early && threadIdx.x == 0u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a.needs_early[4u * b + q] = a.needs_out[4u * b + q];
            __threadfence_system();
            if (__hip_atomic_fetch_add(a.needs_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == a.nb[5])
                __hip_atomic_store(a.needs_bell, a.needs_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#ifndef IVX_POST1_WAVES
#define IVX_POST1_WAVES 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IVX_POST1_WAVES, 8))) void k_step_post1(StepArgs a) { step_post1_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_step_post1_many, StepArgs, step_post1_body, __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IVX_POST1_WAVES, 8))))

// roles: 5 (the launch's FIRST blocks) exact numbering of the multi-region chunks, 0 region merge of multi-region chunks, 1 mesher scan,
// 2 moments final (1 block), 3 occupied final (1 block), 4 the slab protocol's face pairs (component pairs across the upper x face, from the
// neighbour's ids of the exchange before).
// The one dependency inside a launch: the merge of the multi-region chunks reads the labels and region tables the exact numbering writes.
// The numbering's blocks are the first of the grid — workgroups start in index order, and they wait for nothing — and each adds one to
// rscalar[3] (zero at the start of a step, with the other region scalars) behind a release fence when it is through; a merge block that has
// work (rscalar[2] != 0: most steps have no such chunk and nobody waits) spins on that word until all have, bounded, then acquires. The
// numbering needs ~120 registers; this launch is a few hundred blocks of table work and can afford them, k_step_post1 (where the role lived,
// spilling, for a tenth of the speed) cannot.
__device__ __forceinline__ void step_post2_body(const StepArgs& a, uint32_t b, uint32_t) {
    if (b < a.nb[5]) {
        __shared__ CclShared sh;
        if (a.rscalar[2] == 0u) return;  // (no such chunk, as in most steps: nobody waits for this block's word either)
        role_ccl_local_exact(b, a.nb[5], sh, a.flags, a.labels, a.info, a.rparent, a.rscalar, a.multi_list);
        __syncthreads();
        if (threadIdx.x == 0u) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(a.rscalar + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    b -= a.nb[5];
    if (b < a.nb[0]) {
        if (a.nb[5] && __hip_atomic_load(a.rscalar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            if (threadIdx.x == 0u) {
                uint32_t spins = 0;
                while (__hip_atomic_load(a.rscalar + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.nb[5]) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1u << 24)) {  // (seconds: something is badly wrong — flag the step instead of hanging the queue)
                        atomicOr(a.rscalar + 1, 8u);
                        break;
                    }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        role_ccl_merge_multi(b, a.nb[0], a.g, a.labels, a.rparent, a.rscalar, a.multi_list);
        return;
    }
    b -= a.nb[0];
    if (b < a.nb[1]) {
        sn::role_sn_scan(b, a.nb[1], a.n_chunks, a.counts, a.sn_group_sums, a.offsets, a.ranks, a.emit_items, a.walk);
        return;
    }
    b -= a.nb[1];
    if (b < a.nb[2]) {
        role_inertia_final(a.n_partials, a.extent, a.partials, a.moments_out);
        return;
    }
    b -= a.nb[2];
    if (b < a.nb[3]) {
        role_occupied_final(a.n_occ_slots, a.occ_part, a.rscalar + 16);
        return;
    }
    b -= a.nb[3];
    if (b < a.nb[4]) {
        __shared__ uint32_t s_seen[128];
        role_face_pairs(b, a.g, a.fp_side, a.labels, a.rcompid, a.fp_nbr, a.fp_count, a.fp_pairs, a.fp_cap, a.fp_seen, s_seen);
    }
}
__global__ __launch_bounds__(256) void k_step_post2(StepArgs a) { step_post2_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_step_post2_many, StepArgs, step_post2_body, __launch_bounds__(256))

// roles: 0 flatten the region forest, 1 mesher emit, 2 the slab protocol's record
__device__ __forceinline__ void step_emit_body(const StepArgs& a, uint32_t b, uint32_t) {
    if (b < a.nb[0]) {
        role_ccl_flatten(b, a.nb[0], a.g, a.rparent, a.multi_list /* root counts */, a.ccl_group_sums);
        return;
    }
    b -= a.nb[0];
    if (b < a.nb[1]) {
        sn::role_sn_emit<false>(b, a.nb[1], sn_params(a), a.positions, a.normals, a.indices, a.imats, a.submeshes, a.offsets + 2 * (size_t)a.n_chunks + 2,
                                a.emit_items, a.vcap, a.icap, a.scap, nullptr, a.hard_count, a.hard_list, a.hard_count + 32, a.walk, a.n_chunks);
        return;
    }
    b -= a.nb[1];
    if (b < a.nb[2]) {  // (one block) the slab's record for the all-gather; the step's small results where ivx_voxel_step_collect looks for them
        role_step_record(a.rscalar, a.fp_count, a.fp_pairs, a.offsets + 2 * (size_t)a.n_chunks, a.moments_out, a.x_off, a.record_max_pairs, a.record);
        if (a.record_head) {  // (the block's own stores, read back behind a barrier)
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < a.record_head_words; i += 256u) a.record_head[i] = a.record[i];
        }
        if (threadIdx.x < 64u) role_result_gather(a.rscalar, a.offsets + 2 * (size_t)a.n_chunks, a.moments_out, a.work_count, a.eval_count, a.host_block, false, 0u);
    }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_step_emit(StepArgs a) { step_emit_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_step_emit_many, StepArgs, step_emit_body, __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))))

// roles: 0 component ids, 1 the mesher's general pass over the chunks the main pass (k_step_emit) handed on — the launch after the main pass
// anyway; as a launch of its own the general pass cost the step 4 us whether it had a chunk to do or not
__device__ __forceinline__ void step_assign_body(const StepArgs& a, uint32_t b, uint32_t) {
    if (b < a.nb[0]) {
        role_ccl_assign<true>(b, a.nb[0], a.g, a.rparent, a.multi_list /* root offsets inside a group */, a.ccl_group_sums, a.nb[0], a.rcompid, a.rscalar);
        return;
    }
    sn::role_sn_emit_general<false>(b - a.nb[0], a.nb[1], sn_params(a), a.positions, a.normals, a.indices, a.imats, a.vmats, a.submeshes,
                                    a.offsets + 2 * (size_t)a.n_chunks + 2, a.emit_items, a.vcap, a.icap, a.scap, nullptr, a.hard_count, a.hard_list);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_step_assign(StepArgs a) { step_assign_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_step_assign_many, StepArgs, step_assign_body, __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))))

// the results as a launch of their own (one block), ordered after everything enqueued so far
__device__ __forceinline__ void step_gather_body(const StepArgs& a, uint32_t, uint32_t) {
    role_result_gather(a.rscalar, a.offsets + 2 * (size_t)a.n_chunks, a.moments_out, a.work_count, a.eval_count, a.host_block, false, 0u);
    // (the edit path's small results, into pinned host memory: eight loads in flight per lane — one wave copying word by word spent 6 us on 3 KB)
    for (uint32_t i0 = threadIdx.x; i0 < a.copy_words; i0 += 512u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) v[q] = i0 + 64u * q < a.copy_words ? a.copy_src[i0 + 64u * q] : 0u;
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q)
            if (i0 + 64u * q < a.copy_words) a.copy_dst[i0 + 64u * q] = v[q];
    }
    // Doorbell: the sequence number lands after every result word of this (single-wave) block. The stream is in order, so a host that
    // sees it also knows that everything enqueued before this launch is complete (ivx_voxel_step_collect polls it instead of paying
    // the runtime's blocking wait when the step is short).
    __threadfence_system();
    if (threadIdx.x == 0u) __hip_atomic_store(a.host_block + 63, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(64) void k_step_gather(StepArgs a) { step_gather_body(a, 0u, 1u); }
IVX_MANY_TWIN(k_step_gather_many, StepArgs, step_gather_body, __launch_bounds__(64))
IVX_MANY_LAUNCHER(many_post1, k_step_post1_many, StepArgs, 256)
IVX_MANY_LAUNCHER(many_post2, k_step_post2_many, StepArgs, 256)
IVX_MANY_LAUNCHER(many_emit, k_step_emit_many, StepArgs, 256)
IVX_MANY_LAUNCHER(many_assign, k_step_assign_many, StepArgs, 256)
IVX_MANY_LAUNCHER(many_gather, k_step_gather_many, StepArgs, 64)

}  // namespace

static_assert(sizeof(StepArgs) % 8 == 0, "argument blocks travel as 8-byte words");
static const int s_many_registered = (ivx_many_register(IVX_MK_POST1, many_post1, sizeof(StepArgs)), ivx_many_register(IVX_MK_POST2, many_post2, sizeof(StepArgs)),
                                      ivx_many_register(IVX_MK_EMIT, many_emit, sizeof(StepArgs)), ivx_many_register(IVX_MK_ASSIGN, many_assign, sizeof(StepArgs)),
                                      ivx_many_register(IVX_MK_GATHER, many_gather, sizeof(StepArgs)), 0);

static StepArgs make_args(ivx_grid* g) {
    StepArgs a;
    memset(&a, 0, sizeof(a));
    a.g = ivx_view(g);
    a.extent = g->extent;
    a.x_off = g->x_off;
    a.n_chunks = g->n_chunks;
    const uint32_t groups = (g->n_chunks + 255u) / 256u;
    a.flags = g->flags;
    a.labels = g->llabel;
    a.info = g->info;
    a.rparent = g->rparent;
    a.rcompid = g->rcompid;
    a.rscalar = g->rscalar;
    a.multi_list = g->ccl_scratch;
    a.ccl_group_sums = g->group_sums;
    a.touch = g->chunk_touch;
    a.counts = g->chunk_counts;
    a.sn_group_sums = g->group_sums + groups;
    a.offsets = g->chunk_offsets;
    a.ranks = g->chunk_offsets + 2 * (size_t)g->n_chunks + 4;
    a.emit_items = reinterpret_cast<uint4*>(g->sn_list);
    a.work_count = ivx_wc(g);
    a.active_list = g->active_list;
    a.positions = g->positions;
    a.normals = g->normals;
    a.indices = g->indices;
    a.imats = reinterpret_cast<unsigned long long*>(g->index_materials);
    a.vmats = reinterpret_cast<uint4*>(g->vertex_materials);
    a.hard_count = ivx_sn_hard_count(g);
    a.hard_list = g->sn_hard;
    {
        // (entries of at least this many vertices are walked first: above the mean of a smooth body's meshed chunks, ~260; developer knob)
        static const uint32_t big = [] {
            const char* e = getenv("IVX_SN_WALK_BIG");
            return e ? (uint32_t)strtoul(e, nullptr, 10) : 288u;
        }();
        a.walk.items = big ? reinterpret_cast<uint4*>(g->sn_walk) : nullptr;
        a.walk.li = g->sn_walk + 4 * (size_t)g->n_chunks;
        a.walk.count = g->sn_walk + 5 * (size_t)g->n_chunks;  // (zeroed by the count role of k_step_post1)
        a.walk.big = big;
    }
    a.submeshes = g->submeshes;
    a.vcap = (uint32_t)g->vcap;
    a.icap = (uint32_t)g->icap;
    a.scap = (uint32_t)g->scap;
    a.bbox = g->chunk_bbox;
    a.occ_part = g->occ_part;
    a.n_occ_slots = groups;
    a.dens = g->dens_dev;
    a.chunk_moments = g->chunk_moments;
    a.partials = g->partials;
    a.moments_out = g->partials + g->partial_blocks * 10;
    a.host_block = g->result_host_dev;
    a.eval_count = g->samp_len ? g->samp_len + g->n_chunks : nullptr;
    return a;
}

// `stages`: IVX_STAGE_* of this enqueue call (derive already launched when it is part of the call)
int ivx_launch_step_post1(ivx_grid* g, uint32_t stages) {
    if (stages & IVX_STAGE_REMESH)  // (the count role walks the active list)
        if (int rc_l = ivx_ensure_active_list(g)) return rc_l;
    StepArgs a = make_args(g);
    const uint32_t groups = (g->n_chunks + 255u) / 256u;
    if (stages & IVX_STAGE_REMESH) {  // (a wave per run of listed chunks)
        a.count_run = sn::ivx_count_run(g);
        a.nb[0] = (ivx_list_grid(g) + 4u * a.count_run - 1u) / (4u * a.count_run);
    }
    if (stages & IVX_STAGE_REGIONS) a.nb[1] = (g->cc[0] * g->cc[1] + 3u) / 4u;
    if (stages & IVX_STAGE_OCCUPIED) a.nb[3] = groups;
    if (stages & IVX_STAGE_INERTIA) {
        a.nb[4] = groups < (uint32_t)g->partial_blocks ? groups : (uint32_t)g->partial_blocks;
        a.n_partials = a.nb[4];
    }
    if (g->post1_needs_out) {  // (edit path: consumed by this launch)
        for (int d = 0; d < 3; ++d) {
            a.needs_box.t_lo[d] = g->post1_needs_box[d], a.needs_box.t_cc[d] = g->post1_needs_box[3 + d];
            a.needs_box.b_lo[d] = g->post1_needs_box[6 + d], a.needs_box.b_cc[d] = g->post1_needs_box[9 + d];
        }
        a.needs_touched = g->post1_needs_touched;
        a.needs_out = g->post1_needs_out;
        a.needs_early = g->post1_needs_early, a.needs_counter = g->post1_needs_counter, a.needs_bell = g->post1_needs_bell, a.needs_seq = g->post1_needs_seq;
        g->post1_needs_early = nullptr;
        a.nb[5] = a.needs_box.b_cc[0] * a.needs_box.b_cc[1] * a.needs_box.b_cc[2];
        g->post1_needs_out = nullptr;
    }
    if ((stages & IVX_STAGE_REMESH) && g->ghost_event) {
        // a slab whose ghost layers are still on their way (slab_comm.cpp): the count of the chunk planes that read nothing of them as a launch
        // of its own ahead of the wait; the planes beside the ghost layers — and the launch's other roles — behind it
        if (g->ghost_split && ivx_has_interior_planes(g) && a.nb[0]) {
            StepArgs ai = a;
            ai.nb[1] = ai.nb[3] = ai.nb[4] = ai.nb[5] = 0;
            ai.x_part = IVX_XPART_INTERIOR;
            a.x_part = IVX_XPART_FACES;
            if (!ivx_many_try(g->ctx, g, IVX_MK_POST1, ai.nb[0], ai)) IVX_KLAUNCH(k_step_post1, dim3(ai.nb[0]), dim3(256), 0, g->ctx->stream, ai);
        }
        (void)ivx_many_break();
        static const bool skip_wait = getenv("IVX_DEBUG_SKIP_GHOST_WAIT") && atoi(getenv("IVX_DEBUG_SKIP_GHOST_WAIT")) == 2;  // (developer switch, derive.hip)
        if (!skip_wait) IVX_HIP_CHECK(hipStreamWaitEvent(g->ctx->stream, static_cast<hipEvent_t>(g->ghost_event), 0));
        g->ghost_event = nullptr;
    }
    const uint32_t total = a.nb[0] + a.nb[1] + a.nb[3] + a.nb[4] + a.nb[5];
    if (total == 0) return IVX_OK;
    if (!ivx_many_try(g->ctx, g, IVX_MK_POST1, total, a)) IVX_KLAUNCH(k_step_post1, dim3(total), dim3(256), 0, g->ctx->stream, a);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// (slab protocol) the face-pair role's arguments: the neighbour's face ids, count + seen table + pair list in g->pairs_dev
static void face_pair_args(StepArgs& a, ivx_grid* g, const uint16_t* nbr_ids) {
    a.fp_side = 1u;
    a.fp_cap = IVX_MAX_FACE_PAIRS;
    a.fp_nbr = nbr_ids;
    a.fp_count = nbr_ids ? g->pairs_dev : nullptr;
    a.fp_seen = g->pairs_dev + 4;
    a.fp_pairs = reinterpret_cast<uint2*>(g->pairs_dev + 4 + 128);
}

int ivx_launch_step_post2(ivx_grid* g, uint32_t stages, const uint16_t* face_pair_ids) {
    if (g->face_ids_event) {  // (slab protocol: the neighbour's face ids travel apart from its face planes; this launch is their first reader)
        (void)ivx_many_break();
        static const bool skip_wait = getenv("IVX_DEBUG_SKIP_GHOST_WAIT") && atoi(getenv("IVX_DEBUG_SKIP_GHOST_WAIT")) == 3;  // (developer switch, derive.hip)
        if (!skip_wait) IVX_HIP_CHECK(hipStreamWaitEvent(g->ctx->stream, static_cast<hipEvent_t>(g->face_ids_event), 0));
        g->face_ids_event = nullptr;
    }
    StepArgs a = make_args(g);
    if (face_pair_ids) {
        face_pair_args(a, g, face_pair_ids);
        a.nb[4] = (g->cc[1] * g->cc[2] + FACE_COLS - 1u) / FACE_COLS;
    }
    const uint32_t groups = (g->n_chunks + 255u) / 256u;
    if (stages & IVX_STAGE_REGIONS) {
        a.nb[0] = g->n_chunks < 64u ? g->n_chunks : 64u;
        // (the exact numbering: the launch's first blocks, see step_post2_body; a block per sixteen chunks of the grid, 8..128 — a small
        // fragment rarely has more than a handful of such chunks, and with hundreds of objects in one launch idle blocks add up)
        const uint32_t want = g->n_chunks / 16u;
        a.nb[5] = want < 8u ? (g->n_chunks < 8u ? g->n_chunks : 8u) : (want > 128u ? 128u : want);
    }
    if (stages & IVX_STAGE_REMESH) a.nb[1] = groups;
    if (stages & IVX_STAGE_INERTIA) {
        a.nb[2] = 1;
        a.n_partials = groups < (uint32_t)g->partial_blocks ? groups : (uint32_t)g->partial_blocks;
    }
    if (stages & IVX_STAGE_OCCUPIED) a.nb[3] = 1;
    const uint32_t total = a.nb[0] + a.nb[1] + a.nb[2] + a.nb[3] + a.nb[4] + a.nb[5];
    if (total == 0) return IVX_OK;
    if (!ivx_many_try(g->ctx, g, IVX_MK_POST2, total, a)) IVX_KLAUNCH(k_step_post2, dim3(total), dim3(256), 0, g->ctx->stream, a);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_step_emit(ivx_grid* g, uint32_t stages, bool general_in_assign, void* slab_record, bool record_has_pairs) {
    StepArgs a = make_args(g);
    if (stages & IVX_STAGE_REGIONS) a.nb[0] = (g->n_chunks + 255u) / 256u;
    if (stages & IVX_STAGE_REMESH) {
        a.nb[1] = sn::ivx_emit_grid(g, g->n_chunks);
        g->sn_tail_zero = 0;  // (the mesher's counter and cursors: the next incremental remesh clears them itself)
    }
    if (slab_record) {
        face_pair_args(a, g, record_has_pairs ? reinterpret_cast<const uint16_t*>(g->pairs_dev) /* (any non-null value: only the count pointer matters) */ : nullptr);
        a.record = static_cast<unsigned long long*>(slab_record);
        a.record_max_pairs = IVX_MAX_FACE_PAIRS;
        a.record_head = g->record_head_copy;
        a.record_head_words = g->record_head_words;
        a.nb[2] = 1;
    }
    const uint32_t total = a.nb[0] + a.nb[1] + a.nb[2];
    if (total == 0) return IVX_OK;
    if (!ivx_many_try(g->ctx, g, IVX_MK_EMIT, total, a)) IVX_KLAUNCH(k_step_emit, dim3(total), dim3(256), 0, g->ctx->stream, a);
    IVX_HIP_CHECK(hipGetLastError());
    // the chunks the main pass hands on: a role of k_step_assign when that launch follows (the caller says so), else a launch of its own
    if ((stages & IVX_STAGE_REMESH) && !general_in_assign) return ivx_launch_sn_emit_general(g);
    return IVX_OK;
}

// groups of 256 chunks beyond what the fused assign scans in LDS take the stand-alone resolve path (ivx_launch_ccl_resolve)
bool ivx_step_assign_fits(const ivx_grid* g) { return (g->n_chunks + 255u) / 256u <= ASSIGN_MAX_GROUPS; }

// (`with_ccl` false: the mesher's general pass alone, through this launch's twin — the re-emit of many objects whose buffers grew together)
int ivx_launch_step_assign(ivx_grid* g, bool with_mesher_general, bool with_ccl) {
    StepArgs a = make_args(g);
    a.nb[0] = with_ccl ? (g->n_chunks + 255u) / 256u : 0u;
    a.nb[1] = with_mesher_general ? sn::ivx_emit_general_grid(g, g->n_chunks) : 0u;
    if (a.nb[0] + a.nb[1] == 0u) return IVX_OK;
    if (!ivx_many_try(g->ctx, g, IVX_MK_ASSIGN, a.nb[0] + a.nb[1], a)) IVX_KLAUNCH(k_step_assign, dim3(a.nb[0] + a.nb[1]), dim3(256), 0, g->ctx->stream, a);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_step_gather(ivx_grid* g) {
    StepArgs a = make_args(g);
    a.seq = ++g->result_seq;
    a.copy_src = g->gather_copy_src, a.copy_dst = g->gather_copy_dst, a.copy_words = g->gather_copy_words;  // (consumed by this launch)
    g->gather_copy_words = 0;
    if (!ivx_many_try(g->ctx, g, IVX_MK_GATHER, 1u, a)) IVX_KLAUNCH(k_step_gather, dim3(1), dim3(64), 0, g->ctx->stream, a);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

#ifdef IVX_WG_TRACE
// developer build: the probes of the exact numbering as k_step_post2 last ran it (64 chunks x 8 words, chunk & 63)
extern "C" int ivx_debug_exact_trace(unsigned long long* out512) {
    return hipMemcpyFromSymbol(out512, HIP_SYMBOL(ivx_exact_trace_buf), 512 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
