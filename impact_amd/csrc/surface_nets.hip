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
    IVX_HIP_CHECK(hipMemsetAsync(d, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_selftest_division, dim3(257), dim3(256), 0, ctx->stream, d);
    IVX_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_selftest_normalize, dim3(65536), dim3(256), 0, ctx->stream, d);  // 2^28 gradients
    IVX_HIP_CHECK(hipGetLastError());
    IVX_HIP_CHECK(hipMemcpyAsync(mismatches, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    IVX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
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
    const uint32_t groups = (g->n_chunks + 255u) / 256u;
    uint32_t* gs = g->group_sums + groups;  // the first `groups` words belong to the region resolve
    IVX_HIP_CHECK(hipMemsetAsync(gs, 0, sizeof(uint32_t) * (3 * groups + IVX_SN_TAIL_WORDS), g->ctx->stream));  // (stand-alone path; the fused step path presets in its first kernel)
    g->scratch_dirty |= IVX_SCRATCH_SN;
    g->preset_fresh &= ~IVX_SCRATCH_SN;
    const uint32_t run = ivx_count_run(g);
    hipLaunchKernelGGL(k_sn_count, dim3((ivx_list_grid(g) + 4u * run - 1u) / (4u * run)), dim3(256), 0, g->ctx->stream, make_params(g), g->chunk_counts, gs, ivx_wc(g),
                       g->active_list, run);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_sn_scan(ivx_grid* g) {
    // ranks are stored after the offsets/totals block
    hipLaunchKernelGGL(k_sn_scan, dim3((g->n_chunks + 255u) / 256u), dim3(256), 0, g->ctx->stream, g->n_chunks, g->chunk_counts,
                       g->group_sums + (g->n_chunks + 255u) / 256u,
                       g->chunk_offsets, g->chunk_offsets + 2 * (size_t)g->n_chunks + 4, reinterpret_cast<uint4*>(g->sn_list));
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// the counter of the chunks handed to the general pass: the word behind the Surface-Nets group totals (zeroed with them ahead of every count pass);
// the main pass's eight list cursors follow at a stride of 32 words (a cache line each)
uint32_t* ivx_sn_hard_count(ivx_grid* g) { return g->group_sums + 4 * (size_t)((g->n_chunks + 255u) / 256u); }

int ivx_launch_sn_emit(ivx_grid* g) {
    // (stand-alone path, also the re-emit after the buffers grew: the two words may hold an earlier emit's counts; the fused step presets them)
    IVX_HIP_CHECK(hipMemsetAsync(ivx_sn_hard_count(g), 0, IVX_SN_TAIL_WORDS * sizeof(uint32_t), g->ctx->stream));
    const uint32_t blocks = ivx_emit_grid(g, g->n_chunks);
    hipLaunchKernelGGL(k_sn_emit<false>, dim3(blocks), dim3(256), 0, g->ctx->stream, make_params(g), g->positions, g->normals, g->indices,
                       reinterpret_cast<unsigned long long*>(g->index_materials), g->submeshes,
                       g->chunk_offsets + 2 * (size_t)g->n_chunks + 2, reinterpret_cast<const uint4*>(g->sn_list), (uint32_t)g->vcap, (uint32_t)g->icap,
                       (uint32_t)g->scap, nullptr, ivx_sn_hard_count(g), g->sn_hard, ivx_sn_hard_count(g) + 32);
    IVX_HIP_CHECK(hipGetLastError());
    return ivx_launch_sn_emit_general(g);
}

// after the main pass (full remesh): the chunks it handed on
int ivx_launch_sn_emit_general(ivx_grid* g) {
    hipLaunchKernelGGL(k_sn_emit_general<false>, dim3(ivx_emit_general_grid(g, g->n_chunks)), dim3(256), 0, g->ctx->stream, make_params(g), g->positions, g->normals,
                       g->indices, reinterpret_cast<unsigned long long*>(g->index_materials), reinterpret_cast<uint4*>(g->vertex_materials), g->submeshes,
                       g->chunk_offsets + 2 * (size_t)g->n_chunks + 2, reinterpret_cast<const uint4*>(g->sn_list), (uint32_t)g->vcap, (uint32_t)g->icap,
                       (uint32_t)g->scap, nullptr, ivx_sn_hard_count(g), g->sn_hard);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// incremental remesh: emit the listed chunks only — records (chunk, vertex offset, index offset, vertices | quads << 16) and submesh slots come
// from the host-side submesh manager; d_count holds their number
int ivx_launch_sn_emit_list(ivx_grid* g, uint32_t n_records, const uint32_t* d_count, const void* d_records, const uint32_t* d_slots) {
    if (n_records == 0) return IVX_OK;
    IVX_HIP_CHECK(hipMemsetAsync(ivx_sn_hard_count(g), 0, IVX_SN_TAIL_WORDS * sizeof(uint32_t), g->ctx->stream));
    hipLaunchKernelGGL(k_sn_emit<true>, dim3(ivx_emit_grid(g, n_records)), dim3(256), 0, g->ctx->stream, make_params(g), g->positions, g->normals,
                       g->indices, reinterpret_cast<unsigned long long*>(g->index_materials), g->submeshes,
                       d_count, reinterpret_cast<const uint4*>(d_records), (uint32_t)g->vcap, (uint32_t)g->icap, (uint32_t)g->scap, d_slots, ivx_sn_hard_count(g),
                       g->sn_hard, ivx_sn_hard_count(g) + 1);
    IVX_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_sn_emit_general<true>, dim3(ivx_emit_general_grid(g, n_records)), dim3(256), 0, g->ctx->stream, make_params(g), g->positions, g->normals,
                       g->indices, reinterpret_cast<unsigned long long*>(g->index_materials), reinterpret_cast<uint4*>(g->vertex_materials), g->submeshes,
                       d_count, reinterpret_cast<const uint4*>(d_records), (uint32_t)g->vcap, (uint32_t)g->icap, (uint32_t)g->scap, d_slots, ivx_sn_hard_count(g),
                       g->sn_hard);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
