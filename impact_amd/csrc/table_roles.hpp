// Device-side bodies ("roles") of the table-sized passes of a voxel step — occupied ranges, the sum of the chunk moments, the
// step's small results — in the form the fused step launches (step_fused.hip) host them: block index and block count passed in,
// no atomics on words that would need a preset (every block owns a slot; one block combines the slots in the next launch).
#pragma once
#include "ivx_internal.hpp"

namespace ivx_roles {

// update_occupied_ranges (object.rs:1149-1280), first level: min/max of the chunk boxes k_derive / k_chunk_pre left, over the 256
// chunks of block `bid`, into the block's 12-word slot: [0..6) minima (chunk lo xyz, voxel lo xyz), [6..12) maxima.
__device__ __forceinline__ void role_occupied_partial(uint32_t bid, uint32_t cx, uint32_t cy, uint32_t cz, const uint32_t* __restrict__ bbox,
                                                      uint32_t* __restrict__ occ_part) {
    __shared__ uint32_t s_red[4][12];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n = cx * cy * cz;
    const uint32_t c = bid * 256u + tid;
    const uint32_t p = c < n ? bbox[c] : 0u;
    uint32_t v[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) v[q] = q < 6 ? 0xFFFFFFFFu : 0u;
    if (p & 0x80000000u) {
        const uint32_t ck = c % cz, cj = (c / cz) % cy, ci = c / (cz * cy);
        const uint32_t cc[3] = {ci, cj, ck};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            v[d] = cc[d];
            v[6 + d] = cc[d] + 1;
            v[3 + d] = cc[d] * 16u + ((p >> (8 * d)) & 15u);
            v[9 + d] = cc[d] * 16u + ((p >> (8 * d + 4)) & 15u) + 1u;
        }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t t = __shfl_xor(v[q], o, 64);
            v[q] = q < 6 ? min(v[q], t) : max(v[q], t);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 12; ++q) s_red[wave][q] = v[q];
    }
    __syncthreads();
    if (tid < 12u) {
        const uint32_t a = s_red[0][tid], b = s_red[1][tid], c2 = s_red[2][tid], d = s_red[3][tid];
        occ_part[(size_t)bid * 12 + tid] = tid < 6u ? min(min(a, b), min(c2, d)) : max(max(a, b), max(c2, d));
    }
}

// second level: one block combines the slots into raw[12] (plain stores: nothing to preset)
__device__ __forceinline__ void role_occupied_final(uint32_t n_slots, const uint32_t* __restrict__ occ_part, uint32_t* __restrict__ raw) {
    __shared__ uint32_t s_red[4][12];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t v[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) v[q] = q < 6 ? 0xFFFFFFFFu : 0u;
    for (uint32_t s = tid; s < n_slots; s += 256u) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const uint32_t t = occ_part[(size_t)s * 12 + q];
            v[q] = q < 6 ? min(v[q], t) : max(v[q], t);
        }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t t = __shfl_xor(v[q], o, 64);
            v[q] = q < 6 ? min(v[q], t) : max(v[q], t);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 12; ++q) s_red[wave][q] = v[q];
    }
    __syncthreads();
    if (tid < 12u) {
        const uint32_t a = s_red[0][tid], b = s_red[1][tid], c2 = s_red[2][tid], d = s_red[3][tid];
        raw[tid] = tid < 6u ? min(min(a, b), min(c2, d)) : max(max(a, b), max(c2, d));
    }
}

__device__ __forceinline__ double table_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Sum over chunks in chunk order, one THREAD per chunk (inertia.hip, k_inertia_sum): Uniform chunks as closed forms
// (compute_moments_for_uniform_chunk, inertia.rs:703-754), NonUniform chunks from their slots; one partial per block.
__device__ __forceinline__ void role_inertia_sum(uint32_t bid, uint32_t nb, const GridView& g, uint32_t x_off, const float* __restrict__ dens,
                                                 const double* __restrict__ chunk_moments, double* __restrict__ partials) {
    __shared__ double s_red[4][10];
    const uint32_t tid = threadIdx.x;
    const uint32_t n_chunks = g.cx * g.cy * g.cz;
    double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t chunk = bid * 256u + tid; chunk < n_chunks; chunk += nb * 256u) {
        const ivx_chunk_info ci_ = g.info[chunk];
        if (ci_.kind == KIND_UNIFORM) {
            const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
            const double d = (double)dens[ci_.uniform_type];
            const double I0 = (double)((ci + (int)x_off) * 16), J0 = (double)(cj * 16), K0 = (double)(ck * 16);
            const double a1x = 32.0 * I0 + 256.0, a1y = 32.0 * J0 + 256.0, a1z = 32.0 * K0 + 256.0;
            const double I1 = I0 + 16.0, J1 = J0 + 16.0, K1 = K0 + 16.0;
            const double a2x = I1 * I1 * I1 - I0 * I0 * I0, a2y = J1 * J1 * J1 - J0 * J0 * J0, a2z = K1 * K1 * K1 - K0 * K0 * K0;
            s[0] += 4096.0 * d;
            s[1] += 256.0 * d * a1x;
            s[2] += 256.0 * d * a1y;
            s[3] += 256.0 * d * a1z;
            s[4] += 256.0 * d * (a2y + a2z);
            s[5] += 256.0 * d * (a2x + a2z);
            s[6] += 256.0 * d * (a2x + a2y);
            s[7] += 16.0 * d * a1x * a1y;
            s[8] += 16.0 * d * a1y * a1z;
            s[9] += 16.0 * d * a1x * a1z;
        } else if (ci_.kind == KIND_NONUNIFORM) {
#pragma unroll
            for (int m = 0; m < 10; ++m) s[m] += chunk_moments[(size_t)chunk * 10 + m];
        }
    }
    const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
    for (int m = 0; m < 10; ++m) {
        const double v = table_wave_sum(s[m]);
        if (lane == 0) s_red[wave][m] = v;
    }
    __syncthreads();
    if (tid < 10) partials[(size_t)bid * 10 + tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
}

// fixed-order (bitwise reproducible) reduction of the per-block partials by one 256-thread block: wave w sums moments w, w + 4,
// w + 8 — every lane a strided subset in index order, then a fixed shuffle tree (the same order as inertia.hip's k_inertia_final)
__device__ __forceinline__ void role_inertia_final(uint32_t n_blocks, float extent, const double* __restrict__ partials, double* __restrict__ out) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t q = wave; q < 10u; q += 4u) {
        double s = 0.0;
        for (uint32_t b = lane; b < n_blocks; b += 64) s += partials[(size_t)b * 10 + q];
        s = table_wave_sum(s);
        if (lane == 0) {
            const double e = (double)extent, e2 = e * e, e3 = e2 * e, e4 = e2 * e2, e5 = e4 * e;
            const double f = q == 0 ? e3 : (q <= 3 ? 0.5 * e4 : (q <= 6 ? (1.0 / 3.0) * e5 : 0.25 * e5));
            out[q] = s * f;
        }
    }
}

// The small results of a step into the host-mapped block (first 64 threads of the calling block): [0..28) region scalars +
// occupied minima/maxima, [28..31) mesh totals, [31] active chunks, [32..52) the 10 moments (f64 as two words each), [52..55) the
// sampler's three evaluation-list lengths. `region_total_known`: the caller computed word 0 itself in this launch.
__device__ __forceinline__ void role_result_gather(const uint32_t* __restrict__ rscalar, const uint32_t* __restrict__ mesh_totals,
                                                   const double* __restrict__ moments, const uint32_t* __restrict__ work_count,
                                                   const uint32_t* __restrict__ eval_count, uint32_t* __restrict__ host_block, bool region_total_known,
                                                   uint32_t region_total) {
    const uint32_t t = threadIdx.x;
    if (t == 31u) host_block[31] = work_count[0];  // length of the active list (sizes the next step's list-driven grids)
    if (t < 28u) host_block[t] = (t == 0u && region_total_known) ? region_total : rscalar[t];
    if (t < 3u) host_block[28 + t] = mesh_totals[t];
    if (t < 20u) host_block[32 + t] = reinterpret_cast<const uint32_t*>(moments)[t];
    if (t < 3u && eval_count) host_block[52 + t] = eval_count[8u + t];  // (the rolled copy, see role_preset)
}

// Preset of the scratch words a step's stages start from (what k_step_preset did in a launch of its own), as a role of the
// step's first kernel: `gid` = global thread index of that launch (it must have at least max(28, n_sn) threads).
// n_sn = the group totals proper (3 per group of 256 chunks).
struct PresetArgs {
    uint32_t groups;        // IVX_SCRATCH_* bits to preset
    uint32_t* rscalar;      // [0..16) region scalars
    uint32_t* sn_sums;      // [n_sn] Surface-Nets group totals + list counter
    uint32_t n_sn;
    uint32_t* eval_count;   // [5] sampler evaluation list counters (may be null)
};
__device__ __forceinline__ void role_preset(const PresetArgs& a, uint32_t gid) {
    if ((a.groups & IVX_SCRATCH_REGIONS) && gid < 16u) a.rscalar[gid] = 0u;
    if ((a.groups & IVX_SCRATCH_SN) && gid < a.n_sn) a.sn_sums[gid] = 0u;
    // behind the group totals: the general mesher pass's counter and the main pass's eight list cursors, a cache line apart (IVX_SN_TAIL_WORDS)
    if ((a.groups & IVX_SCRATCH_SN) && gid < 9u) a.sn_sums[a.n_sn + 32u * gid] = 0u;
    if ((a.groups & IVX_SCRATCH_EVAL) && gid < 5u && a.eval_count) a.eval_count[gid] = 0u;
    // the sampler's list counters rolled over after their last reader (k_sdf_eval): kept as statistics in words [8..13), zero for the
    // next step's pre-pass — the step after needs no kernel ahead of the pre-pass just to clear five words
    if ((a.groups & IVX_SCRATCH_EVAL_ROLL) && gid < 5u && a.eval_count) {
        a.eval_count[8u + gid] = a.eval_count[gid];
        a.eval_count[gid] = 0u;
    }
}

}  // namespace ivx_roles
