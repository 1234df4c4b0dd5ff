// a5-a7 — padded chunk SDF + Surface Nets + mesh assembly (full rebuild), one workgroup per exposed
// chunk. Three launches: count -> scan (chunk-linear exclusive offsets) -> emit.
//
// Reference behaviour reproduced (engine/crates/impact_voxel/src/):
//   for_each_exposed_chunk_with_sdf / padding fills     object/sdf.rs:156-508
//   compute_surface_nets_mesh                           object/sdf/surface_nets.rs:131-148
//   estimate_surface_nets_surface (vertex scan order)   surface_nets.rs:152-244
//   centroid_of_edge_intersections, CUBE_EDGES          surface_nets.rs:384-418, 661-674
//   trilinear gradient                                  object/sdf.rs:603-633
//   make_all_surface_nets_quads / diagonal split        surface_nets.rs:251-381
//   SurfaceNetsVertexMaterials::compute + sort network  surface_nets.rs:428-522
//   calculate_index_materials_for_triangle              surface_nets.rs:559-637
//   VoxelObjectMesh::recreate (concatenation, offsets)  mesh.rs:286-354, 559-577
//   ChunkSubmesh obscuredness table                     mesh.rs:611-635
//
// CDNA4 mapping: the 18^3 padded tile (i8 distance + u8 type) is staged in LDS (11.4 KiB); the
// reference's sequential (i,j,k) vertex order is reproduced by ordered stream compaction — 64-lane
// ballot + popcount prefix inside each wave, wave totals through LDS, a running base across the 20
// passes of 256 cubes. Quads are compacted the same way over vertices (<=3 quads each, X then Y then Z).
// Exact f32 arithmetic (no FMA, IEEE div/sqrt) keeps the diagonal choice and hence the index buffer
// bit-identical to the reference.
#include "sn_roles.hpp"
#include "many.hpp"

namespace {
using namespace ivx_roles::sn;

// stand-alone forms of the roles in sn_roles.hpp
__global__ __launch_bounds__(256) void k_sn_count(SnParams p, uint32_t* __restrict__ counts, uint32_t* __restrict__ group_sums,
                                                  const uint32_t* __restrict__ work_counts, const uint32_t* __restrict__ active_list, uint32_t run) {
    __shared__ uint32_t s_rows[4 * NROWS];
    role_sn_count_waves(blockIdx.x, gridDim.x, p, counts, group_sums, work_counts, active_list, s_rows, run);
}
__global__ __launch_bounds__(256) void k_sn_scan(uint32_t n_chunks, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ group_sums,
                                                 uint32_t* __restrict__ offsets, uint32_t* __restrict__ ranks, uint4* __restrict__ emit_items) {
    role_sn_scan(blockIdx.x, gridDim.x, n_chunks, counts, group_sums, offsets, ranks, emit_items);
}
// (amdgpu_waves_per_eu(4): keeps the kernel at <= 128 VGPRs so that four workgroups fit a CU; the LDS footprint, ~38 KB, allows
// four as well. The kernel is bound by the latency of a workgroup's serial phases, so residency is what buys throughput.)
template <bool SLOTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sn_emit(SnParams p, float* __restrict__ positions, float* __restrict__ normals,
                                                 uint32_t* __restrict__ indices, unsigned long long* __restrict__ imats, ivx_submesh* __restrict__ submeshes,
                                                 const uint32_t* __restrict__ emit_count, const uint4* __restrict__ emit_items, uint32_t vcap, uint32_t icap,
                                                 uint32_t scap, const uint32_t* __restrict__ slots, uint32_t* __restrict__ hard_count,
                                                 uint32_t* __restrict__ hard_list, uint32_t* __restrict__ cursor) {
    role_sn_emit<SLOTS>(blockIdx.x, gridDim.x, p, positions, normals, indices, imats, submeshes, emit_count, emit_items, vcap, icap, scap, slots, hard_count, hard_list,
                        cursor);
}
// the chunks the main pass handed on (several materials around a vertex or a quad, more vertices than its LDS cache holds)
template <bool SLOTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sn_emit_general(SnParams p, float* __restrict__ positions, float* __restrict__ normals,
                                                 uint32_t* __restrict__ indices, unsigned long long* __restrict__ imats,
                                                 uint4* __restrict__ vmats, ivx_submesh* __restrict__ submeshes, const uint32_t* __restrict__ emit_count,
                                                 const uint4* __restrict__ emit_items, uint32_t vcap, uint32_t icap, uint32_t scap,
                                                 const uint32_t* __restrict__ slots, const uint32_t* __restrict__ hard_count,
                                                 const uint32_t* __restrict__ hard_list) {
    role_sn_emit_general<SLOTS>(blockIdx.x, gridDim.x, p, positions, normals, indices, imats, vmats, submeshes, emit_count, emit_items, vcap, icap, scap, slots,
                                hard_count, hard_list);
}

__global__ __launch_bounds__(256) void k_box_mesh_needs(SnParams p, BoxNeeds bx, const uint32_t* __restrict__ touched, const uint32_t* __restrict__ list,
                                                        uint32_t* __restrict__ out) {
    __shared__ uint32_t s_lds[NROWS + 2];
    role_box_mesh_needs(blockIdx.x, p, bx, touched, list, out, s_lds);
}

// the incremental remesh's two launches (ivx_launch_sn_emit_list) with their arguments as one block each: the twins of the many-object path
struct SnEmitArgs {
    SnParams p;
    float* positions;
    float* normals;
    uint32_t* indices;
    unsigned long long* imats;
    uint4* vmats;
    ivx_submesh* submeshes;
    const uint32_t* emit_count;
    const uint4* emit_items;
    const uint32_t* slots;
    uint32_t* hard_count;
    uint32_t* hard_list;
    uint32_t* cursor;
    uint32_t vcap, icap, scap, n_patch;
    // (incremental remesh) entries of the submesh table that a removal moved: `n_patch` of them, written to their slots by the main pass's first
    // block — five small copies of their own cost the host 20 us
    const ivx_submesh* patch_entries;
    const uint32_t* patch_slots;
};
__device__ __forceinline__ void sn_emit_slots_body(const SnEmitArgs& a, uint32_t bid, uint32_t nb) {
    if (bid == 0u && a.n_patch) {
        constexpr uint32_t W = (uint32_t)(sizeof(ivx_submesh) / sizeof(uint32_t));
        for (uint32_t i = threadIdx.x; i < a.n_patch * W; i += 256u)
            reinterpret_cast<uint32_t*>(a.submeshes + a.patch_slots[i / W])[i % W] = reinterpret_cast<const uint32_t*>(a.patch_entries)[i];
    }
    role_sn_emit<true>(bid, nb, a.p, a.positions, a.normals, a.indices, a.imats, a.submeshes, a.emit_count, a.emit_items, a.vcap, a.icap, a.scap, a.slots, a.hard_count,
                       a.hard_list, a.cursor);
}
__device__ __forceinline__ void sn_emit_general_slots_body(const SnEmitArgs& a, uint32_t bid, uint32_t nb) {
    role_sn_emit_general<true>(bid, nb, a.p, a.positions, a.normals, a.indices, a.imats, a.vmats, a.submeshes, a.emit_count, a.emit_items, a.vcap, a.icap, a.scap, a.slots,
                               a.hard_count, a.hard_list);
    // The last workgroup through leaves the counter, the main pass's list cursors and its own census word zero for the next incremental
    // remesh (every workgroup has read the counter by then): that one's host does not enqueue a fill of 288 words ahead of its main pass.
    __shared__ uint32_t s_last;
    __syncthreads();
    if (threadIdx.x == 0u) {
        __threadfence();
        s_last = atomicAdd(a.hard_count + (IVX_SN_TAIL_WORDS - 1u), 1u) == nb - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (s_last)
        for (uint32_t w = threadIdx.x; w < IVX_SN_TAIL_WORDS; w += 256u) a.hard_count[w] = 0u;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sn_emit_slots(SnEmitArgs a) { sn_emit_slots_body(a, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sn_emit_general_slots(SnEmitArgs a) {
    sn_emit_general_slots_body(a, blockIdx.x, gridDim.x);
}
IVX_MANY_TWIN(k_sn_emit_slots_many, SnEmitArgs, sn_emit_slots_body, __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))))
IVX_MANY_TWIN(k_sn_emit_general_slots_many, SnEmitArgs, sn_emit_general_slots_body, __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))))
IVX_MANY_LAUNCHER(many_sn_emit_slots, k_sn_emit_slots_many, SnEmitArgs, 256)
IVX_MANY_LAUNCHER(many_sn_emit_general_slots, k_sn_emit_general_slots_many, SnEmitArgs, 256)
static_assert(sizeof(SnEmitArgs) % 8 == 0, "argument blocks travel as 8-byte words");
static const int s_many_registered_sn = (ivx_many_register(IVX_MK_SN_EMIT_SLOTS, many_sn_emit_slots, sizeof(SnEmitArgs)),
                                         ivx_many_register(IVX_MK_SN_EMIT_GENERAL_SLOTS, many_sn_emit_general_slots, sizeof(SnEmitArgs)), 0);

// div_ranged against the `/` operator over the whole operand set the mesher hands it: t = d1 / (d1 - d2) for every pair of decoded
// distances of opposite sign (surface_nets.rs:396-404; decoded 0 is +0.0 and counts as positive), and 1 / n for the edge counts.
__global__ __launch_bounds__(256) void k_selftest_division(uint32_t* __restrict__ mismatches) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;  // 65 536 pairs + 256 counts
    uint32_t bad = 0u;
    if (t < 65536u) {
        const int8_t e1 = (int8_t)(t & 0xFFu), e2 = (int8_t)(t >> 8);
        if ((e1 < 0) != (e2 < 0)) {
            const float d1 = decode(e1), d2 = decode(e2);
            const float a = d1 / (d1 - d2), b = div_ranged(d1, d1 - d2);
            bad = __float_as_uint(a) != __float_as_uint(b) ? 1u : 0u;
        }
    } else {
        const float n = (float)(t - 65536u + 1u);
        bad = __float_as_uint(1.0f / n) != __float_as_uint(div_ranged(1.0f, n)) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, 1u);
}

// normalize_ranged against sqrtf and `/` over pseudo-random gradients: components with random signs and mantissas, exponents spread over
// 2^-40 .. 2^8 (one component in eight an exact zero), 16 per thread; triples outside the function's domain (len2 < 2^-100) are skipped as
// the mesher skips them.
__global__ __launch_bounds__(256) void k_selftest_normalize(uint32_t* __restrict__ mismatches) {
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 0x9E3779B9u + 0x7F4A7C15u;
    auto next = [&]() {
        h ^= h << 13;
        h ^= h >> 17;
        h ^= h << 5;
        return h;
    };
    uint32_t bad = 0u;
    for (int it = 0; it < 16; ++it) {
        float g[3];
        const uint32_t spread = next();
        for (int c = 0; c < 3; ++c) {
            const uint32_t r = next(), r2 = next();
            const uint32_t ex = 127u - 40u + (r2 % 49u);
            g[c] = (r2 >> 29) == 0u ? 0.0f : __uint_as_float((r & 0x807FFFFFu) | (ex << 23));
            if (spread & (1u << c)) g[c] = __uint_as_float((r & 0x807FFFFFu) | ((120u + ((r2 >> 8) & 7u)) << 23));  // all of one magnitude: the usual case
        }
        const float len2 = (g[0] * g[0] + g[1] * g[1]) + g[2] * g[2];
        if (!(len2 >= 0x1p-100f)) continue;
        float nx, ny, nz;
        normalize_ranged(g[0], g[1], g[2], len2, nx, ny, nz);
        const float gl = sqrtf(len2);
        if (__float_as_uint(nx) != __float_as_uint(g[0] / gl) || __float_as_uint(ny) != __float_as_uint(g[1] / gl) ||
            __float_as_uint(nz) != __float_as_uint(g[2] / gl))
            bad += 1u;
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace

// (test hook, include/impact_voxel_hip.h)
int ivx_selftest_mesher_division(ivx_ctx* ctx, uint32_t* mismatches) {
    IVX_REQUIRE(ctx && mismatches, IVX_ERR_INVALID, "ivx_selftest_mesher_division: null argument");
    uint32_t* d = nullptr;
    IVX_HIP_CHECK(hipMalloc(&d, sizeof(uint32_t)));
    IVX_HIP_CHECK(ivx_memset_async(d, 0, sizeof(uint32_t), ctx->stream));
    IVX_KLAUNCH(k_selftest_division, dim3(257), dim3(256), 0, ctx->stream, d);
    IVX_HIP_CHECK(hipGetLastError());
    IVX_KLAUNCH(k_selftest_normalize, dim3(65536), dim3(256), 0, ctx->stream, d);  // 2^28 gradients
    IVX_HIP_CHECK(hipGetLastError());
    IVX_HIP_CHECK(ivx_memcpy_async(mismatches, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    IVX_HIP_CHECK(ivx_stream_sync(ctx->stream));
    IVX_HIP_CHECK(hipFree(d));
    return IVX_OK;
}

static SnParams make_params(ivx_grid* g) {
    SnParams p;
    p.g = ivx_view(g);
    p.extent = g->extent;
    p.x_off = g->x_off;
    return p;
}

int ivx_launch_sn_count(ivx_grid* g) {
    if (int rc_l = ivx_ensure_active_list(g)) return rc_l;
    const uint32_t groups = (g->n_chunks + 255u) / 256u;
    uint32_t* gs = g->group_sums + groups;  // the first `groups` words belong to the region resolve
    IVX_HIP_CHECK(ivx_memset_async(gs, 0, sizeof(uint32_t) * (IVX_SN_GROUP_WORDS * groups + IVX_SN_TAIL_WORDS), g->ctx->stream));  // (stand-alone path; the fused step path presets in its first kernel)
    g->scratch_dirty |= IVX_SCRATCH_SN;
    g->preset_fresh &= ~IVX_SCRATCH_SN;
    const uint32_t run = ivx_count_run(g);
    IVX_KLAUNCH(k_sn_count, dim3((ivx_list_grid(g) + 4u * run - 1u) / (4u * run)), dim3(256), 0, g->ctx->stream, make_params(g), g->chunk_counts, gs, ivx_wc(g),
                       g->active_list, run);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_box_mesh_needs(ivx_grid* g, const uint32_t t_lo[3], const uint32_t t_cc[3], const uint32_t b_lo[3], const uint32_t b_cc[3], const uint32_t* d_touched,
                              uint32_t* d_out) {
    BoxNeeds bx;
    for (int d = 0; d < 3; ++d) bx.t_lo[d] = t_lo[d], bx.t_cc[d] = t_cc[d], bx.b_lo[d] = b_lo[d], bx.b_cc[d] = b_cc[d];
    const uint32_t n = b_cc[0] * b_cc[1] * b_cc[2];
    if (n == 0) return IVX_OK;
    IVX_KLAUNCH(k_box_mesh_needs, dim3(n), dim3(256), 0, g->ctx->stream, make_params(g), bx, d_touched, nullptr, d_out);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
int ivx_launch_list_mesh_needs(ivx_grid* g, uint32_t n, const uint32_t* d_list, uint32_t* d_out) {
    if (n == 0) return IVX_OK;
    BoxNeeds bx;
    memset(&bx, 0, sizeof(bx));
    bx.b_cc[0] = bx.b_cc[1] = bx.b_cc[2] = 1u;  // (unused divisors)
    IVX_KLAUNCH(k_box_mesh_needs, dim3(n), dim3(256), 0, g->ctx->stream, make_params(g), bx, nullptr, d_list, d_out);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_sn_scan(ivx_grid* g) {
    // ranks are stored after the offsets/totals block
    IVX_KLAUNCH(k_sn_scan, dim3((g->n_chunks + 255u) / 256u), dim3(256), 0, g->ctx->stream, g->n_chunks, g->chunk_counts,
                       g->group_sums + (g->n_chunks + 255u) / 256u,
                       g->chunk_offsets, g->chunk_offsets + 2 * (size_t)g->n_chunks + 4, reinterpret_cast<uint4*>(g->sn_list));
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// the counter of the chunks handed to the general pass: the word behind the Surface-Nets group totals (zeroed with them ahead of every count pass);
// the main pass's eight list cursors follow at a stride of 32 words (a cache line each)
uint32_t* ivx_sn_hard_count(ivx_grid* g) { return g->group_sums + (1u + IVX_SN_GROUP_WORDS) * (size_t)((g->n_chunks + 255u) / 256u); }

int ivx_launch_sn_emit(ivx_grid* g) {
    g->sn_tail_zero = 0;
    // (stand-alone path, also the re-emit after the buffers grew: the two words may hold an earlier emit's counts; the fused step presets them)
    IVX_HIP_CHECK(ivx_memset_async(ivx_sn_hard_count(g), 0, IVX_SN_TAIL_WORDS * sizeof(uint32_t), g->ctx->stream));
    const uint32_t blocks = ivx_emit_grid(g, g->n_chunks);
    IVX_KLAUNCH(k_sn_emit<false>, dim3(blocks), dim3(256), 0, g->ctx->stream, make_params(g), g->positions, g->normals, g->indices,
                       reinterpret_cast<unsigned long long*>(g->index_materials), g->submeshes,
                       g->chunk_offsets + 2 * (size_t)g->n_chunks + 2, reinterpret_cast<const uint4*>(g->sn_list), (uint32_t)g->vcap, (uint32_t)g->icap,
                       (uint32_t)g->scap, nullptr, ivx_sn_hard_count(g), g->sn_hard, ivx_sn_hard_count(g) + 32);
    IVX_HIP_CHECK(hipGetLastError());
    return ivx_launch_sn_emit_general(g);
}

// after the main pass (full remesh): the chunks it handed on
int ivx_launch_sn_emit_general(ivx_grid* g) {
    IVX_KLAUNCH(k_sn_emit_general<false>, dim3(ivx_emit_general_grid(g, g->n_chunks)), dim3(256), 0, g->ctx->stream, make_params(g), g->positions, g->normals,
                       g->indices, reinterpret_cast<unsigned long long*>(g->index_materials), reinterpret_cast<uint4*>(g->vertex_materials), g->submeshes,
                       g->chunk_offsets + 2 * (size_t)g->n_chunks + 2, reinterpret_cast<const uint4*>(g->sn_list), (uint32_t)g->vcap, (uint32_t)g->icap,
                       (uint32_t)g->scap, nullptr, ivx_sn_hard_count(g), g->sn_hard);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// incremental remesh: emit the listed chunks only — records (chunk, vertex offset, index offset, vertices | quads << 16) and submesh slots come
// from the host-side submesh manager; d_count holds their number
int ivx_launch_sn_emit_list(ivx_grid* g, uint32_t n_records, const uint32_t* d_count, const void* d_records, const uint32_t* d_slots, uint32_t n_patch,
                            const void* d_patch_entries, const uint32_t* d_patch_slots) {
    if (n_records == 0) return IVX_OK;
    // (the general pass of the incremental remesh before this one left the words zero, unless a step's mesher has used them since)
    if (!g->sn_tail_zero && !ivx_many_zero(g->ctx, g, ivx_sn_hard_count(g), IVX_SN_TAIL_WORDS * sizeof(uint32_t))) {
        (void)ivx_many_break();
        IVX_HIP_CHECK(ivx_memset_async(ivx_sn_hard_count(g), 0, IVX_SN_TAIL_WORDS * sizeof(uint32_t), g->ctx->stream));
    }
    g->sn_tail_zero = 1;
    SnEmitArgs a;
    memset(&a, 0, sizeof(a));
    a.p = make_params(g), a.positions = g->positions, a.normals = g->normals, a.indices = g->indices;
    a.imats = reinterpret_cast<unsigned long long*>(g->index_materials), a.vmats = reinterpret_cast<uint4*>(g->vertex_materials), a.submeshes = g->submeshes;
    a.emit_count = d_count, a.emit_items = reinterpret_cast<const uint4*>(d_records), a.slots = d_slots, a.hard_count = ivx_sn_hard_count(g), a.hard_list = g->sn_hard;
    a.cursor = ivx_sn_hard_count(g) + 1, a.vcap = (uint32_t)g->vcap, a.icap = (uint32_t)g->icap, a.scap = (uint32_t)g->scap;
    a.n_patch = n_patch, a.patch_entries = static_cast<const ivx_submesh*>(d_patch_entries), a.patch_slots = d_patch_slots;
    const uint32_t b_main = ivx_emit_grid(g, n_records), b_gen = ivx_emit_general_grid(g, n_records);
    if (!ivx_many_try(g->ctx, g, IVX_MK_SN_EMIT_SLOTS, b_main, a)) IVX_KLAUNCH(k_sn_emit_slots, dim3(b_main), dim3(256), 0, g->ctx->stream, a);
    IVX_HIP_CHECK(hipGetLastError());
    if (!ivx_many_try(g->ctx, g, IVX_MK_SN_EMIT_GENERAL_SLOTS, b_gen, a)) IVX_KLAUNCH(k_sn_emit_general_slots, dim3(b_gen), dim3(256), 0, g->ctx->stream, a);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
