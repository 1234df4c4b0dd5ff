// a9-a11 — connected regions ("split detection"): two-level connected-component labelling that mirrors
// the reference's structure (engine/crates/impact_voxel/src/object/split_detection.rs:15-61):
//   level 1  chunk-local regions, u8 label per voxel (255 = empty), <=254 regions per chunk,
//            boundary-touching regions numbered first      split_detection.rs:662-891
//   level 2  a disjoint-set forest over (chunk, local region) nodes joined across chunk faces
//                                                          split_detection.rs:323-487, 1046-1460, 1914-1974
//   count_regions / find_two_disconnected_regions          split_detection.rs:193-301
//
// The reference's raw label VALUES are artefacts of its sequential union order, so the
// contract checked by the tests is the partition: canonicalised labels bit-equal to the oracle's, equal
// per-chunk (region_count, boundary_region_count) and equal count_regions.
//
// CDNA4 mapping: level 1 runs entirely in LDS (16 KiB parent array per workgroup): each thread links
// the runs of its 16-voxel row in registers, then joins rows across +x/+y with lock-free atomicMin
// unions (root = smallest voxel index, so labels are deterministic); region numbers come from two
// ordered ballot/prefix compactions (boundary roots, then interior roots). Level 2 uses global
// atomicMin unions over the small (chunk,region) table; voxels never carry a 32-bit label in HBM.
// Traffic: level 1 reads 1 B/voxel (flags) and writes 1 B/voxel (label); level 2 reads only the six
// face planes of labels per chunk.
#include "ccl_roles.hpp"
#include "table_roles.hpp"

namespace {
using namespace ivx_roles;

// Level 1 as a kernel of its own (the step path runs it fused into k_derive, see chunk_passes.hpp): walks the active list;
// the non-empty masks come from the flags plane.
__global__ __launch_bounds__(256) void k_ccl_local(GridView g, const uint8_t* __restrict__ flags, uint8_t* __restrict__ labels,
                                                   ivx_chunk_info* __restrict__ info, uint32_t* __restrict__ rparent,
                                                   uint32_t* __restrict__ rscalar, uint32_t* __restrict__ multi_list,
                                                   const uint32_t* __restrict__ work_counts, const uint32_t* __restrict__ active_list) {
    __shared__ CclShared sh;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_active = work_counts[0];
    for (uint32_t li = ivx_xcd_remap(blockIdx.x, gridDim.x); li < n_active; li += gridDim.x) {
        __syncthreads();  // the previous chunk's LDS use is over
        const uint32_t entry = active_list[li];  // chunk + kinds (written by k_derive): no trip to the chunk record
        const uint32_t chunk = IVX_LIST_CHUNK(entry);
        const uint32_t m = flags_mask(*reinterpret_cast<const uint4*>(flags + (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)tid * 16));
        uint32_t rc, brc;
        ccl_local_chunk(sh, tid, chunk, IVX_LIST_KIND(entry), IVX_LIST_GEN(entry), m, labels, rparent, rscalar, multi_list, rc, brc);
        if (tid == 0) {
            info[chunk].region_count = (uint8_t)rc;
            info[chunk].boundary_region_count = (uint8_t)brc;
        }
    }
}

// The kernels below are the stand-alone forms of the roles in ccl_roles.hpp (the step path runs the same roles inside the
// fused launches of step_fused.hip).
__global__ __launch_bounds__(256) void k_ccl_local_exact(const uint8_t* __restrict__ flags, uint8_t* __restrict__ labels, ivx_chunk_info* __restrict__ info,
                                                         uint32_t* __restrict__ rparent, uint32_t* __restrict__ rscalar, const uint32_t* __restrict__ multi_list) {
    __shared__ CclShared sh;
    role_ccl_local_exact(blockIdx.x, gridDim.x, sh, flags, labels, info, rparent, rscalar, multi_list);
}
__global__ __launch_bounds__(256) void k_ccl_merge_columns(GridView g, const uint8_t* __restrict__ touch, uint32_t* __restrict__ rparent) {
    role_ccl_merge_columns(blockIdx.x, gridDim.x, g, touch, rparent);
}
__global__ __launch_bounds__(256) void k_ccl_merge_multi(GridView g, const uint8_t* __restrict__ labels, uint32_t* __restrict__ rparent,
                                                         const uint32_t* __restrict__ rscalar, const uint32_t* __restrict__ multi_list) {
    role_ccl_merge_multi(blockIdx.x, gridDim.x, g, labels, rparent, rscalar, multi_list);
}
__global__ __launch_bounds__(256) void k_ccl_flatten(GridView g, uint32_t* __restrict__ rparent, uint32_t* __restrict__ root_counts,
                                                     uint32_t* __restrict__ group_sums) {
    role_ccl_flatten(blockIdx.x, gridDim.x, g, rparent, root_counts, group_sums);
}
__global__ __launch_bounds__(256) void k_scan_groups(uint32_t n, const uint32_t* __restrict__ in, const uint32_t* __restrict__ group_sums,
                                                     uint32_t* __restrict__ out, uint32_t* __restrict__ total) {
    role_scan_groups(blockIdx.x, gridDim.x, n, in, group_sums, out, total);
}
template <bool FUSED>
__global__ __launch_bounds__(256) void k_ccl_assign(GridView g, const uint32_t* __restrict__ rparent, const uint32_t* __restrict__ root_offsets,
                                                    const uint32_t* __restrict__ group_sums, uint32_t n_groups, uint32_t* __restrict__ rcompid,
                                                    uint32_t* __restrict__ total) {
    role_ccl_assign<FUSED>(blockIdx.x, gridDim.x, g, rparent, root_offsets, group_sums, n_groups, rcompid, total);
}

__global__ __launch_bounds__(256) void k_ccl_dense(uint32_t n_chunks, const uint8_t* __restrict__ labels, const uint32_t* __restrict__ rcompid,
                                                   uint32_t* __restrict__ out) {
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = blockIdx.x;
    const size_t base = (size_t)chunk * IVX_CHUNK_VOXELS;
    for (int it = 0; it < 16; ++it) {
        const uint32_t idx = it * 256 + tid;
        const uint32_t l = labels[base + idx];
        out[base + idx] = l == 255u ? NODE_NONE : rcompid[chunk * 256u + l];
    }
}


// ---- cross-slab (multi-GPU) region exchange: component id of every voxel of an x face, and the
// distinct (own component, neighbour component) pairs across that face. Together with an all-gather
// of the pairs this is the cross-chunk connection step of the reference's global resolve
// (split_detection.rs:323-487, 1046-1325) carried across ranks.
// slab-local component id of a face voxel as 16 bits (0xFFFF = empty): what crosses the link per face voxel. A slab with
// 65 535 or more components cannot be described this way: error bit 2 in rscalar[1].
__device__ __forceinline__ uint32_t face_id16(const GridView& g, uint32_t side, uint32_t col, uint32_t tid, const uint8_t* labels, const uint32_t* rcompid,
                                              uint32_t* rscalar) {
    const uint32_t chunk = (side ? g.cx - 1 : 0u) * g.cy * g.cz + col;
    const uint32_t ckind = g.info[chunk].kind;
    const uint32_t l = ckind != KIND_NONUNIFORM ? ivx_uniform_label(ckind) : labels[(size_t)chunk * IVX_CHUNK_VOXELS + ((side ? 15u : 0u) << 8) + tid];
    if (l == 255u) return 0xFFFFu;
    const uint32_t id = rcompid[chunk * 256u + l];
    if (id >= 0xFFFFu) atomicOr(&rscalar[1], 4u);
    return id & 0xFFFFu;
}
__global__ __launch_bounds__(256) void k_face_ids(GridView g, uint32_t side, const uint8_t* __restrict__ labels,
                                                  const uint32_t* __restrict__ rcompid, uint32_t* __restrict__ rscalar, uint16_t* __restrict__ out) {
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = (uint16_t)face_id16(g, side, blockIdx.x, threadIdx.x, labels, rcompid, rscalar);
}

// Both faces of a slab in one launch (blockIdx.y = side): the one-voxel face planes of (sdf, type) + the face layer's chunk
// records, and — after the region stages — the component ids of the face voxels right behind them. A chunk that is only its
// record (compact planes) is expanded here. A wave per chunk column, four face voxels a lane (a face plane of a chunk is 256 consecutive
// bytes of its planes): words in, words out — a thread per voxel moved single bytes and the launch was 6.5-9.5 us of every rank's step, twice.
// with_ids: 0 the planes and records, 1 both, 2 the ids alone (the planes went ahead, slab_comm.cpp)
__global__ __launch_bounds__(256) void k_halo_pack_both(GridView g, uint8_t* __restrict__ out_lo, uint8_t* __restrict__ out_hi, uint32_t with_ids,
                                                        const uint8_t* __restrict__ labels, const uint32_t* __restrict__ rcompid,
                                                        uint32_t* __restrict__ rscalar, uint32_t* __restrict__ pair_words) {
    const uint32_t side = blockIdx.y, tid = threadIdx.x, lane = tid & 63u;
    const size_t cols = (size_t)g.cy * g.cz;
    const uint32_t col = blockIdx.x * 4u + (tid >> 6);
    // (the face-pair count and seen table of the pass that follows the exchange start at zero: cleared here instead of by a fill of their own)
    if (pair_words && blockIdx.x == 0 && side == 0 && tid < 4u + 128u) pair_words[tid] = 0u;
    uint8_t* out = side ? out_hi : out_lo;
    if (!out || col >= cols) return;
    const uint32_t chunk = (side ? g.cx - 1 : 0u) * g.cy * g.cz + col;
    const size_t src = (size_t)chunk * IVX_CHUNK_VOXELS + ((side ? 15u : 0u) << 8) + 4u * lane;
    const ivx_chunk_info rec = g.info[chunk];
    const bool dense = rec.kind == KIND_NONUNIFORM;
    if (!(with_ids & 2u)) {
        const uint32_t sd = dense ? *reinterpret_cast<const uint32_t*>(g.sdf + src) : ((uint32_t)ivx_uniform_sdf(rec.kind) & 0xFFu) * 0x01010101u;
        const uint32_t ty = dense ? *reinterpret_cast<const uint32_t*>(g.type + src) : ((uint32_t)ivx_uniform_type(rec) & 0xFFu) * 0x01010101u;
        *reinterpret_cast<uint32_t*>(out + (size_t)col * 256 + 4u * lane) = sd;
        *reinterpret_cast<uint32_t*>(out + cols * 256 + (size_t)col * 256 + 4u * lane) = ty;
        if (lane == 0) reinterpret_cast<ivx_chunk_info*>(out + cols * 512)[col] = rec;
    }
    if (with_ids) {
        // slab-local component ids of the four face voxels as 16 bits each (face_id16)
        const uint32_t lw = dense ? *reinterpret_cast<const uint32_t*>(labels + src) : ivx_uniform_label(rec.kind) * 0x01010101u;
        uint32_t id[4];
        bool over = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t l = (lw >> (8 * q)) & 0xFFu;
            id[q] = 0xFFFFu;
            if (l != 255u) {
                const uint32_t v = rcompid[chunk * 256u + l];
                over = over || v >= 0xFFFFu;
                id[q] = v & 0xFFFFu;
            }
        }
        if (over) atomicOr(&rscalar[1], 4u);
        *reinterpret_cast<uint2*>(out + cols * (512 + sizeof(ivx_chunk_info)) + ((size_t)col * 256 + 4u * lane) * 2) = make_uint2(id[0] | (id[1] << 16), id[2] | (id[3] << 16));
    }
}

// (bodies: role_face_pairs / role_step_record in ccl_roles.hpp — the slab protocol's remesh phase hosts them in the step's fused launches)
__global__ __launch_bounds__(256) void k_face_pairs(GridView g, uint32_t side, const uint8_t* __restrict__ labels,
                                                    const uint32_t* __restrict__ rcompid, const uint16_t* __restrict__ nbr,
                                                    uint32_t* __restrict__ n_pairs, uint2* __restrict__ pairs, uint32_t cap, uint32_t* __restrict__ seen) {
    __shared__ uint32_t s_seen[128];
    role_face_pairs(blockIdx.x, g, side, labels, rcompid, nbr, n_pairs, pairs, cap, seen, s_seen);
}

__global__ __launch_bounds__(256) void k_step_record(const uint32_t* __restrict__ rscalar, const uint32_t* __restrict__ pair_count,
                                                     const uint2* __restrict__ pairs, const uint32_t* __restrict__ mesh_totals,
                                                     const double* __restrict__ moments, uint32_t x_off, uint32_t max_pairs,
                                                     unsigned long long* __restrict__ rec, const uint32_t* __restrict__ work_count,
                                                     const uint32_t* __restrict__ eval_count, uint32_t* __restrict__ host_block) {
    role_step_record(rscalar, pair_count, pairs, mesh_totals, moments, x_off, max_pairs, rec);
    // (with `host_block`: the step's small results where ivx_voxel_step_collect looks for them, so that it needs no launch of its own)
    if (host_block && threadIdx.x < 64u) role_result_gather(rscalar, mesh_totals, moments, work_count, eval_count, host_block, false, 0u);
}

// ---- per-region statistics (what extract_disconnected_region needs to pick and size a fragment,
// object/extraction.rs:121-295, plus the moments the PropertyTransferrer would move, inertia.rs:341-560)
__device__ __forceinline__ double wsum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wmin_u(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t wmax_u(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_down(v, o, 64));
    return v;
}

struct RegionStatsOut {
    unsigned long long* count;  // [n]
    uint32_t* lo;               // [3n] init 0xFFFFFFFF
    uint32_t* hi;               // [3n] init 0
    uint32_t* nu_chunks;        // [n]
    uint32_t* chunks;           // [n]
    uint32_t* root;             // [n]
    double* moments;            // [10n] integer-form sums (scaled on the host)
};

__global__ __launch_bounds__(256) void k_region_stats(GridView g, uint32_t x_off, const uint8_t* __restrict__ labels,
                                                      const uint32_t* __restrict__ rparent, const uint32_t* __restrict__ rcompid,
                                                      const float* __restrict__ dens, RegionStatsOut out) {
    __shared__ float s_dens[256];
    __shared__ double s_m[4][10];
    __shared__ uint32_t s_u[4][7];
    const uint32_t tid = threadIdx.x;
    const uint32_t n_chunks = g.cx * g.cy * g.cz;
    const uint32_t chunk = ivx_xcd_remap(blockIdx.x, n_chunks);
    const ivx_chunk_info info = g.info[chunk];
    const uint32_t rc = info.region_count;
    if (rc == 0) return;
    s_dens[tid] = dens[tid];
    const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
    const int ti = tid >> 4, tj = tid & 15;
    const size_t o = (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)tid * 16;
    const uint4 l4 = *reinterpret_cast<const uint4*>(labels + o);
    const uint4 t4 = *reinterpret_cast<const uint4*>(g.type + o);
    const uint32_t lw[4] = {l4.x, l4.y, l4.z, l4.w}, tw[4] = {t4.x, t4.y, t4.z, t4.w};
    const uint32_t gi = (uint32_t)(ci + (int)x_off) * 16u + ti, gj = cj * 16u + tj;
    const double I = (double)gi, J = (double)gj;
    const double qx = 2.0 * I + 1.0, qy = 2.0 * J + 1.0;
    const double cx = 3.0 * I * I + 3.0 * I + 1.0, cy = 3.0 * J * J + 3.0 * J + 1.0;
    __syncthreads();
    for (uint32_t r = 0; r < rc; ++r) {
        uint32_t mask = 0;
        double D = 0.0, Dz1 = 0.0, Dz2 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (((lw[k >> 2] >> (8 * (k & 3))) & 0xFFu) == r) {
                mask |= 1u << k;
                const double d = (double)s_dens[(tw[k >> 2] >> (8 * (k & 3))) & 0xFFu];
                const double K = (double)(ck * 16 + k);
                D += d;
                Dz1 += d * (2.0 * K + 1.0);
                Dz2 += d * (3.0 * K * K + 3.0 * K + 1.0);
            }
        }
        double s[10] = {D, D * qx, D * qy, Dz1, D * cy + Dz2, D * cx + Dz2, D * (cx + cy), D * qx * qy, qy * Dz1, qx * Dz1};
        uint32_t u[7];
        u[0] = __popc(mask);
        u[1] = mask ? gi : 0xFFFFFFFFu;
        u[2] = mask ? gj : 0xFFFFFFFFu;
        u[3] = mask ? (uint32_t)(ck * 16 + __ffs(mask) - 1) : 0xFFFFFFFFu;
        u[4] = mask ? gi + 1 : 0u;
        u[5] = mask ? gj + 1 : 0u;
        u[6] = mask ? (uint32_t)(ck * 16 + 32 - __clz(mask)) : 0u;
        const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            double v = wsum_d(s[q]);
            if (lane == 0) s_m[wave][q] = v;
        }
        {
            uint32_t v = u[0];
#pragma unroll
            for (int oo = 32; oo > 0; oo >>= 1) v += __shfl_down(v, oo, 64);
            if (lane == 0) s_u[wave][0] = v;
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                uint32_t w = wmin_u(u[q]);
                if (lane == 0) s_u[wave][q] = w;
            }
#pragma unroll
            for (int q = 4; q < 7; ++q) {
                uint32_t w = wmax_u(u[q]);
                if (lane == 0) s_u[wave][q] = w;
            }
        }
        __syncthreads();
        const uint32_t node = chunk * 256u + r;
        const uint32_t comp = rcompid[node];
        if (comp != NODE_NONE) {
            if (tid < 10) atomicAdd(&out.moments[(size_t)comp * 10 + tid], ((s_m[0][tid] + s_m[1][tid]) + s_m[2][tid]) + s_m[3][tid]);
            if (tid == 10) atomicAdd(&out.count[comp], (unsigned long long)(s_u[0][0] + s_u[1][0] + s_u[2][0] + s_u[3][0]));
            if (tid >= 11 && tid < 14) {
                const int q = tid - 10;
                atomicMin(&out.lo[(size_t)comp * 3 + (q - 1)], min(min(s_u[0][q], s_u[1][q]), min(s_u[2][q], s_u[3][q])));
            }
            if (tid >= 14 && tid < 17) {
                const int q = tid - 10;
                atomicMax(&out.hi[(size_t)comp * 3 + (q - 4)], max(max(s_u[0][q], s_u[1][q]), max(s_u[2][q], s_u[3][q])));
            }
            if (tid == 17) {
                // a chunk counts once per component even when several of its local regions belong to it
                // (`found_region` in extraction.rs:163-205)
                bool first = true;
                for (uint32_t q = 0; q < r; ++q) first = first && rcompid[chunk * 256u + q] != comp;
                if (first) {
                    atomicAdd(&out.chunks[comp], 1u);
                    if (info.kind == KIND_NONUNIFORM) atomicAdd(&out.nu_chunks[comp], 1u);
                }
                if (rparent[node] == node) out.root[comp] = node;
            }
        }
        __syncthreads();
    }
}

}  // namespace

int ivx_launch_ccl_local(ivx_grid* g, int fused) {
    if (int rc_l = ivx_ensure_active_list(g)) return rc_l;
    GridView v = ivx_view(g);
    if (!fused) IVX_HIP_CHECK(ivx_memset_async(g->rscalar, 0, 16 * sizeof(uint32_t), g->ctx->stream));
    g->scratch_dirty |= IVX_SCRATCH_REGIONS;
    uint32_t* multi_list = g->ccl_scratch;  // reused by the resolve pass afterwards
    g->planes_compact = 1;
    if (!fused)  // else k_derive labelled the chunks in the same sweep
        IVX_KLAUNCH(k_ccl_local, dim3(ivx_list_grid(g)), dim3(256), 0, g->ctx->stream, v, g->flags, g->llabel, g->info, g->rparent, g->rscalar,
                           multi_list, ivx_wc(g), g->active_list);
    const uint32_t exact_blocks = g->n_chunks < 1024u ? g->n_chunks : 1024u;
    IVX_KLAUNCH(k_ccl_local_exact, dim3(exact_blocks), dim3(256), 0, g->ctx->stream, g->flags, g->llabel, g->info, g->rparent, g->rscalar, multi_list);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_ccl_local_only(ivx_grid* g) {
    if (int rc_l = ivx_ensure_active_list(g)) return rc_l;
    GridView v = ivx_view(g);
    g->planes_compact = 1;
    g->scratch_dirty |= IVX_SCRATCH_REGIONS;
    IVX_KLAUNCH(k_ccl_local, dim3(ivx_list_grid(g)), dim3(256), 0, g->ctx->stream, v, g->flags, g->llabel, g->info, g->rparent, g->rscalar, g->ccl_scratch,
                       ivx_wc(g), g->active_list);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_ccl_merge(ivx_grid* g) {
    GridView v = ivx_view(g);
    IVX_KLAUNCH(k_ccl_merge_columns, dim3((g->cc[0] * g->cc[1] + 3u) / 4u), dim3(256), 0, g->ctx->stream, v, g->chunk_touch, g->rparent);
    const uint32_t multi_blocks = g->n_chunks < 256u ? g->n_chunks : 256u;
    IVX_KLAUNCH(k_ccl_merge_multi, dim3(multi_blocks), dim3(256), 0, g->ctx->stream, v, g->llabel, g->rparent, g->rscalar, g->ccl_scratch);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_ccl_resolve(ivx_grid* g) {
    GridView v = ivx_view(g);
    uint32_t* root_counts = g->ccl_scratch;
    uint32_t* root_offsets = g->ccl_scratch + g->n_chunks;
    const uint32_t nb = (g->n_chunks + 255u) / 256u;
    uint32_t* group_sums = g->group_sums;
    IVX_KLAUNCH(k_ccl_flatten, dim3(nb), dim3(256), 0, g->ctx->stream, v, g->rparent, root_counts, group_sums);
    if (nb <= ASSIGN_MAX_GROUPS) {
        IVX_KLAUNCH(k_ccl_assign<true>, dim3(nb), dim3(256), 0, g->ctx->stream, v, g->rparent, root_counts, group_sums, nb, g->rcompid, g->rscalar);
    } else {
        IVX_KLAUNCH(k_scan_groups, dim3(nb), dim3(256), 0, g->ctx->stream, g->n_chunks, root_counts, group_sums, root_offsets, g->rscalar);
        IVX_KLAUNCH(k_ccl_assign<false>, dim3(nb), dim3(256), 0, g->ctx->stream, v, g->rparent, root_offsets, group_sums, nb, g->rcompid, g->rscalar);
    }
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_ccl_dense_labels(ivx_grid* g, uint32_t* d_labels) {
    {
        int rc = ivx_ensure_dense(g);
        if (rc) return rc;
    }
    IVX_KLAUNCH(k_ccl_dense, dim3(g->n_chunks), dim3(256), 0, g->ctx->stream, g->n_chunks, g->llabel, g->rcompid, d_labels);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_halo_pack_both(ivx_grid* g, void* buf_lo, void* buf_hi, int with_face_labels) {
    GridView v = ivx_view(g);
    IVX_KLAUNCH(k_halo_pack_both, dim3((g->cc[1] * g->cc[2] + 3u) / 4u, 2), dim3(256), 0, g->ctx->stream, v, static_cast<uint8_t*>(buf_lo),
                       static_cast<uint8_t*>(buf_hi), with_face_labels ? 1u : 0u, g->llabel, g->rcompid, g->rscalar,
                       (with_face_labels && g->pairs_dev) ? g->pairs_dev : nullptr);
    IVX_HIP_CHECK(hipGetLastError());
    g->pairs_zeroed = (with_face_labels && g->pairs_dev) ? 1 : 0;
    return IVX_OK;
}

int ivx_launch_halo_pack_parts(ivx_grid* g, void* buf_lo, void* buf_hi, uint32_t what) {
    if (what == 3u) return ivx_launch_halo_pack_both(g, buf_lo, buf_hi, 1);
    if (what == 1u) return ivx_launch_halo_pack_both(g, buf_lo, buf_hi, 0);
    GridView v = ivx_view(g);
    IVX_KLAUNCH(k_halo_pack_both, dim3((g->cc[1] * g->cc[2] + 3u) / 4u, 2), dim3(256), 0, g->ctx->stream, v, static_cast<uint8_t*>(buf_lo), static_cast<uint8_t*>(buf_hi), 2u, g->llabel,
                g->rcompid, g->rscalar, g->pairs_dev ? g->pairs_dev : nullptr);
    IVX_HIP_CHECK(hipGetLastError());
    g->pairs_zeroed = g->pairs_dev ? 1 : 0;
    return IVX_OK;
}

int ivx_launch_face_ids(ivx_grid* g, int side, uint16_t* d_out) {
    GridView v = ivx_view(g);
    IVX_KLAUNCH(k_face_ids, dim3(g->cc[1] * g->cc[2]), dim3(256), 0, g->ctx->stream, v, (uint32_t)side, g->llabel, g->rcompid, g->rscalar, d_out);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_face_pairs(ivx_grid* g, int side, const uint16_t* d_nbr, uint32_t* d_count, void* d_pairs, uint32_t cap, uint32_t* d_seen) {
    GridView v = ivx_view(g);
    if (d_seen == d_count + 4) {  // count and seen-table are neighbours (the enqueue path): one fill, unless the pack kernel before cleared them
        if (!(g->pairs_zeroed && d_count == g->pairs_dev)) IVX_HIP_CHECK(ivx_memset_async(d_count, 0, (4 + 128) * sizeof(uint32_t), g->ctx->stream));
        g->pairs_zeroed = 0;
    } else {
        IVX_HIP_CHECK(ivx_memset_async(d_count, 0, sizeof(uint32_t), g->ctx->stream));
        if (d_seen) IVX_HIP_CHECK(ivx_memset_async(d_seen, 0, 128 * sizeof(uint32_t), g->ctx->stream));
    }
    IVX_KLAUNCH(k_face_pairs, dim3((g->cc[1] * g->cc[2] + FACE_COLS - 1u) / FACE_COLS), dim3(256), 0, g->ctx->stream, v, (uint32_t)side, g->llabel, g->rcompid, d_nbr,
                       d_count, static_cast<uint2*>(d_pairs), cap, d_seen);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_step_record(ivx_grid* g, const uint32_t* d_pair_count, const void* d_pairs, uint32_t max_pairs, void* d_record, bool with_results) {
    IVX_KLAUNCH(k_step_record, dim3(1), dim3(256), 0, g->ctx->stream, g->rscalar, d_pair_count, static_cast<const uint2*>(d_pairs),
                       g->chunk_offsets + 2 * (size_t)g->n_chunks, g->partials + g->partial_blocks * 10, g->x_off, max_pairs,
                       static_cast<unsigned long long*>(d_record), ivx_wc(g), g->samp_len ? g->samp_len + g->n_chunks : nullptr,
                       with_results ? g->result_host_dev : nullptr);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_region_stats(ivx_grid* g, const float* d_dens, void* d_buf, uint32_t n) {
    // layout of d_buf: count u64[n] | moments f64[10n] | lo u32[3n] | hi u32[3n] | nu u32[n] | chunks u32[n] | root u32[n]
    RegionStatsOut o;
    char* p = static_cast<char*>(d_buf);
    o.count = reinterpret_cast<unsigned long long*>(p);
    p += sizeof(unsigned long long) * n;
    o.moments = reinterpret_cast<double*>(p);
    p += sizeof(double) * 10 * n;
    o.lo = reinterpret_cast<uint32_t*>(p);
    p += sizeof(uint32_t) * 3 * n;
    o.hi = reinterpret_cast<uint32_t*>(p);
    p += sizeof(uint32_t) * 3 * n;
    o.nu_chunks = reinterpret_cast<uint32_t*>(p);
    p += sizeof(uint32_t) * n;
    o.chunks = reinterpret_cast<uint32_t*>(p);
    p += sizeof(uint32_t) * n;
    o.root = reinterpret_cast<uint32_t*>(p);
    const size_t head = (sizeof(unsigned long long) + sizeof(double) * 10) * n;
    IVX_HIP_CHECK(ivx_memset_async(d_buf, 0, head, g->ctx->stream));
    IVX_HIP_CHECK(ivx_memset_async(o.lo, 0xFF, sizeof(uint32_t) * 3 * n, g->ctx->stream));
    IVX_HIP_CHECK(ivx_memset_async(o.hi, 0, sizeof(uint32_t) * (3 + 1 + 1 + 1) * n, g->ctx->stream));
    GridView v = ivx_view(g);
    {
        int rc = ivx_ensure_dense(g);
        if (rc) return rc;
    }
    IVX_KLAUNCH(k_region_stats, dim3(g->n_chunks), dim3(256), 0, g->ctx->stream, v, g->x_off, g->llabel, g->rparent, g->rcompid, d_dens, o);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
