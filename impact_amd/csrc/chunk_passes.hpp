// Per-chunk passes shared by the stand-alone kernels (ccl.hip, inertia.hip) and the fused sweep in derive.hip: the
// chunk-local connected regions and the chunk's moments need nothing but the chunk's own non-empty row masks (and types),
// which k_derive holds in registers anyway.
#pragma once
#include "ivx_internal.hpp"

#define NODE_NONE 0xFFFFFFFFu

__device__ __forceinline__ uint32_t lds_find(volatile uint32_t* par, uint32_t x) {
    uint32_t p;
    while ((p = par[x]) != x) x = p;
    return x;
}
__device__ __forceinline__ void lds_union(uint32_t* par, uint32_t a, uint32_t b) {
    for (int guard = 0; guard < 8192; ++guard) {
        a = lds_find(par, a);
        b = lds_find(par, b);
        if (a == b) return;
        if (a < b) {
            uint32_t t = a;
            a = b;
            b = t;
        }
        uint32_t old = atomicMin(&par[a], b);  // attach the larger root under the smaller
        if (old == a) return;
        a = old;
    }
}


struct CclShared {
    uint32_t par[IVX_CHUNK_VOXELS];
    uint32_t mask[256];
    uint32_t cnt;
};

// Level 1 of the region labelling for one chunk (split_detection.rs:662-891), all 256 threads of the workgroup: decides
// whether the chunk holds 0, 1 or several regions. One region (the overwhelmingly common case) needs no numbering: label 0
// on every non-empty voxel. Chunks with several regions go on a list for k_ccl_local_exact, which owns their labels, counts
// and region table. Union-find over the RUNS of non-empty voxels along k (a thread owns the <= 8 runs of its 16-voxel row);
// the node of a run is the voxel index of its first voxel, links go through LDS atomicMin (root = smallest index).
// `m`: the thread's non-empty row mask (ignored when the chunk is Void or was generated Uniform). Writes the label plane
// (NonUniform chunks only: compact planes) and the first slot of the region table; returns the region counts for the
// caller to put into the chunk record (same values in every thread).
__device__ __forceinline__ void ccl_local_chunk(CclShared& sh, uint32_t tid, uint32_t chunk, uint32_t kind, uint32_t gen, uint32_t m_in,
                                                uint8_t* __restrict__ labels, uint32_t* __restrict__ rparent, uint32_t* __restrict__ rscalar,
                                                uint32_t* __restrict__ multi_list, uint32_t& rc_out, uint32_t& brc_out) {
    const int ti = tid >> 4, tj = tid & 15;
    const size_t base = (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)tid * 16;
    uint32_t* rp = rparent + (size_t)chunk * 256;
    // a Void chunk has no voxels and a chunk generated Uniform is one solid region whether or not it was demoted since
    const bool known = kind == KIND_VOID || gen == KIND_UNIFORM;
    const uint32_t m = known ? 0u : m_in;
    const int all_full = known ? (kind != KIND_VOID) : __syncthreads_and(m == 0xFFFFu);
    const int any = known ? (kind != KIND_VOID) : __syncthreads_or(m != 0);
    if (!any || all_full) {
        // no voxels, or one solid region touching every face
        const uint32_t lab = any ? 0u : 0xFFFFFFFFu;
        if (kind == KIND_NONUNIFORM) *reinterpret_cast<uint4*>(labels + base) = make_uint4(lab, lab, lab, lab);  // else: compact planes
        // only slots below region_count are ever read (flatten / assign / find walk valid nodes only)
        if (tid == 0) rp[0] = any ? chunk * 256u : NODE_NONE;
        rc_out = any ? 1u : 0u;
        brc_out = any ? 1u : 0u;
        return;
    }
    // 1. one node per run, keyed by the voxel index of its first voxel
    sh.mask[tid] = m;
    const uint32_t starts = m & ~(m << 1);
    {
        uint32_t r = starts;
        while (r) {
            const int k = __ffs(r) - 1;
            r &= r - 1;
            sh.par[tid * 16 + k] = tid * 16 + k;
        }
    }
    if (tid == 0) sh.cnt = 0;
    __syncthreads();
    // 2. join runs across +x and +y: one union per run of the overlap between the two rows
    {
        const uint32_t mx = ti < 15 ? sh.mask[tid + 16] : 0u;
        const uint32_t my = tj < 15 ? sh.mask[tid + 1] : 0u;
        const uint32_t sx = mx & ~(mx << 1), sy = my & ~(my << 1);
        uint32_t bx = m & mx, by = m & my;
        bx &= ~(bx << 1);
        by &= ~(by << 1);
        while (bx) {
            const int k = __ffs(bx) - 1;
            bx &= bx - 1;
            const uint32_t lowk = (2u << k) - 1u;  // bits 0..k
            const uint32_t a = tid * 16 + (31 - __clz(starts & lowk)), b = (tid + 16) * 16 + (31 - __clz(sx & lowk));
            lds_union(sh.par, a, b);
        }
        while (by) {
            const int k = __ffs(by) - 1;
            by &= by - 1;
            const uint32_t lowk = (2u << k) - 1u;
            const uint32_t a = tid * 16 + (31 - __clz(starts & lowk)), b = (tid + 1) * 16 + (31 - __clz(sy & lowk));
            lds_union(sh.par, a, b);
        }
    }
    __syncthreads();
    // 3. count the roots (a root keeps itself as parent, every other node points somewhere else, so no flattening is needed
    // to count); does any voxel lie on the chunk boundary?
    uint32_t n_roots = 0;
    {
        uint32_t r = starts;
        while (r) {
            const int k = __ffs(r) - 1;
            r &= r - 1;
            n_roots += sh.par[tid * 16 + k] == tid * 16 + (uint32_t)k;
        }
    }
    {
        const uint32_t wr = ivx_wave_sum(n_roots);
        if ((tid & 63u) == 0 && wr) atomicAdd(&sh.cnt, wr);
    }
    const bool edge_row = ti == 0 || ti == 15 || tj == 0 || tj == 15;
    const int touches = __syncthreads_or(edge_row ? (m != 0) : ((m & 0x8001u) != 0));
    const uint32_t rc = sh.cnt;
    if (rc == 1u) {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (!((m >> k) & 1u)) w[k >> 2] |= 0xFFu << (8 * (k & 3));
        *reinterpret_cast<uint4*>(labels + base) = make_uint4(w[0], w[1], w[2], w[3]);
        if (tid == 0) rp[0] = chunk * 256u;
        rc_out = 1u;
        brc_out = touches ? 1u : 0u;
    } else {
        // several regions: the reference's numbering is reproduced by k_ccl_local_exact (which also sets the boundary count)
        if (tid == 0) multi_list[atomicAdd(&rscalar[2], 1u)] = chunk;
        rc_out = rc < 254u ? rc : 254u;
        brc_out = 0u;
    }
}

// ---- moments of one NonUniform chunk (inertia.rs:615-699 in the integer form described in inertia.hip) ---------------------
// Sum over the wave on the VALU's DPP path (a shuffle goes through the LDS crossbar, two ds_bpermute per step and double; ten
// of those chains per chunk made the moment pass wait on LDS for a third of its time). Rows of 16 lanes are summed with row_shr
// 1/2/4/8 (lanes shifted in from outside the row read zero), lane 15 of each row then holds the row total; the four row totals
// are read as scalars and added in a fixed order. Every lane returns the wave total.
template <int CTRL>
__device__ __forceinline__ double ivx_dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ivx_wave_sum_f64(double v) {
    v += ivx_dpp_f64<0x111>(v);  // row_shr:1
    v += ivx_dpp_f64<0x112>(v);  // row_shr:2
    v += ivx_dpp_f64<0x114>(v);  // row_shr:4
    v += ivx_dpp_f64<0x118>(v);  // row_shr:8
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * q + 15), __builtin_amdgcn_readlane(lo, 16 * q + 15));
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// all 256 threads; `m` non-empty mask of the thread's row, `tw` its 16 type bytes, (gi, gj) the row's global voxel indices,
// k0 the chunk's first k; s_dens the 256 densities and s_red[4][10] scratch in LDS. Ends with the ten sums in `out10`.
__device__ __forceinline__ void chunk_moments_rows(uint32_t tid, uint32_t m, const uint32_t tw[4], const float* s_dens, double (*s_red)[10], int gi, int gj,
                                                   int k0, double* __restrict__ out10) {
    const double I = (double)gi, J = (double)gj;
    const double qx = 2.0 * I + 1.0, qy = 2.0 * J + 1.0;
    const double cx = 3.0 * I * I + 3.0 * I + 1.0, cy = 3.0 * J * J + 3.0 * J + 1.0;
    double D = 0.0, Dz1 = 0.0, Dz2 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if ((m >> k) & 1u) {
            const double d = (double)s_dens[(tw[k >> 2] >> (8 * (k & 3))) & 0xFFu];
            const double K = (double)(k0 + k);
            D += d;
            Dz1 += d * (2.0 * K + 1.0);
            Dz2 += d * (3.0 * K * K + 3.0 * K + 1.0);
        }
    }
    const double s[10] = {D, D * qx, D * qy, Dz1, D * cy + Dz2, D * cx + Dz2, D * (cx + cy), D * qx * qy, qy * Dz1, qx * Dz1};
    const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const double v = ivx_wave_sum_f64(s[q]);
        if (lane == 0) s_red[wave][q] = v;
    }
    __syncthreads();
    if (tid < 10) out10[tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
}
