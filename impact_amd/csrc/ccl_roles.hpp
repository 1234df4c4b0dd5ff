// Device-side bodies ("roles") of the region kernels, shared by ccl.hip's stand-alone kernels and the fused step launches
// (step_fused.hip): a role is what one workgroup of the original kernel does, with its block index and block count passed in
// (`bid`, `nb`), so that one launch can host several independent roles side by side. Algorithm notes and reference citations:
// ccl.hip.
#pragma once
#include "chunk_passes.hpp"

namespace ivx_roles {

__device__ __forceinline__ uint32_t flags_mask(uint4 f) {
    uint32_t w[4] = {f.x, f.y, f.z, f.w};
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (!((w[k >> 2] >> (8 * (k & 3))) & VF_EMPTY)) m |= 1u << k;
    return m;
}

__device__ __forceinline__ uint32_t prefix_ordered(uint32_t val, uint32_t* s_wsum, uint32_t tid, uint32_t& total) {
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t incl = val;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o) incl += n;
    }
    if (lane == 63u) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t w0 = s_wsum[0], w1 = s_wsum[1], w2 = s_wsum[2], w3 = s_wsum[3];
    uint32_t wbase = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
    total = w0 + w1 + w2 + w3;
    __syncthreads();
    return wbase + incl - val;
}


// Exact chunk-local numbering of the chunks with several regions (the list ccl_local_chunk made): one workgroup per listed chunk,
// all 256 threads (ccl_exact_chunk, chunk_passes.hpp). The non-empty masks come from the flags plane.
__device__ __forceinline__ void role_ccl_local_exact(uint32_t bid, uint32_t nb, CclShared& sh, const uint8_t* __restrict__ flags, uint8_t* __restrict__ labels,
                                                     ivx_chunk_info* __restrict__ info, uint32_t* __restrict__ rparent, uint32_t* __restrict__ rscalar,
                                                     const uint32_t* __restrict__ multi_list) {
    const uint32_t tid = threadIdx.x;
    const uint32_t n_multi = rscalar[2];
    for (uint32_t li = bid; li < n_multi; li += nb) {  // bounded grid-stride walk over the (usually empty) list
        const uint32_t chunk = multi_list[li];
        __syncthreads();
        sh.mask[tid] = flags_mask(*reinterpret_cast<const uint4*>(flags + (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)tid * 16));
        __syncthreads();
        uint32_t rc, brc;
        ccl_exact_chunk(sh, tid, chunk, labels, rparent, rscalar, rc, brc);
        if (tid == 0) {
            info[chunk].region_count = (uint8_t)rc;
            info[chunk].boundary_region_count = (uint8_t)brc;
        }
    }
}

// ---- level 2 ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t g_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// find with path halving: a node's parent only ever moves to an ancestor, and ancestors have smaller indices (unions hang the
// larger root under the smaller), so shortening with atomicMin is safe beside concurrent unions and other finds
__device__ __forceinline__ uint32_t g_find(uint32_t* par, uint32_t x) {
    uint32_t p = g_load(par + x);
    while (p != x) {
        const uint32_t gp = g_load(par + p);
        if (gp != p) atomicMin(par + x, gp);
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void g_union(uint32_t* par, uint32_t a, uint32_t b) {
    for (int guard = 0; guard < (1 << 20); ++guard) {
        a = g_find(par, a);
        b = g_find(par, b);
        if (a == b) return;
        if (a < b) {
            uint32_t t = a;
            a = b;
            b = t;
        }
        uint32_t old = atomicMin(par + a, b);
        if (old == a) return;
        a = old;
    }
}

// Level 2 for single-region chunks (all but a handful). Joining chunks pairwise through the forest costs a chain of dependent
// global loads and atomics per pair, nearly all of them on the one root of the body. They are joined by structure instead:
// one WAVE per (ci, cj) column of chunks, lane = ck. Two single-region chunks stacked along k are linked when k_derive saw a
// voxel pair touch across their face; a run of linked chunks hangs directly under its first chunk (plain stores: those
// nodes are nobody's root yet). Where a chunk touches its +y / +x neighbour, the heads of the two runs are joined, once
// per stretch over which both runs continue. Chunks with several regions are left to k_ccl_merge_multi.
__device__ __forceinline__ void role_ccl_merge_columns(uint32_t bid, uint32_t nb, GridView g, const uint8_t* __restrict__ touch, uint32_t* __restrict__ rparent) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t col = bid * 4u + (threadIdx.x >> 6);
    if (col >= g.cx * g.cy) return;
    const uint32_t cj = col % g.cy, ci = col / g.cy;
    // heads of the runs that reach the end of the previous 64-chunk segment: this column, the +y and the +x column
    uint32_t carry[3] = {NODE_NONE, NODE_NONE, NODE_NONE};
    const uint32_t cols[3] = {col, col + 1u, col + g.cy};
    const bool has[3] = {true, cj + 1 < g.cy, ci + 1 < g.cx};
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t k0 = 0; k0 < g.cz; k0 += 64u) {
        const uint32_t ck = k0 + lane;
        const bool in = ck < g.cz;
        bool single[3];
        unsigned long long lm[3];
        uint32_t head[3];
        uint32_t own_touch = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            single[q] = false;
            bool linkz = false;
            if (in && has[q]) {
                const uint32_t c = cols[q] * g.cz + ck;
                const uint32_t tb = touch[c];
                single[q] = g.info[c].region_count == 1;
                if (q == 0) own_touch = tb;
                if (single[q] && ck + 1 < g.cz && ((tb >> 2) & 1u)) linkz = g.info[c + 1].region_count == 1;
            }
            lm[q] = __ballot(linkz);
            // node of the first chunk of the run of column q that contains this lane (runs continue across segments)
            const unsigned long long gaps = ~lm[q] & below;
            const uint32_t hs = gaps ? 64u - (uint32_t)__clzll(gaps) : 0u;
            head[q] = (hs == 0u && carry[q] != NODE_NONE) ? carry[q] : (cols[q] * g.cz + k0 + hs) * 256u;
        }
        const uint32_t node = (col * g.cz + ck) * 256u;
        if (single[0] && head[0] != node) rparent[node] = head[0];
#pragma unroll
        for (int q = 1; q < 3; ++q) {
            const bool link = single[0] && single[q] && ((own_touch >> (q == 1 ? 1 : 0)) & 1u);
            const unsigned long long links = __ballot(link);
            // the lane below joined the same two runs already
            const bool dup = lane != 0 && ((links >> (lane - 1)) & 1ull) && ((lm[0] >> (lane - 1)) & 1ull) && ((lm[q] >> (lane - 1)) & 1ull);
            if (link && !dup) g_union(rparent, head[0], head[q]);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const uint32_t h63 = __shfl(head[q], 63, 64);
            carry[q] = ((lm[q] >> 63) & 1ull) ? h63 : NODE_NONE;
        }
    }
}

// Chunks with several regions (the list k_ccl_local made): one workgroup per such chunk joins its regions with those of
// all SIX neighbours through the label planes (a single-region neighbour does not look at this pair itself).
__device__ __forceinline__ void role_ccl_merge_multi(uint32_t bid, uint32_t nb, GridView g, const uint8_t* __restrict__ labels, uint32_t* __restrict__ rparent,
                                                         const uint32_t* __restrict__ rscalar, const uint32_t* __restrict__ multi_list) {
    const uint32_t tid = threadIdx.x;
    const uint32_t n_multi = rscalar[2];
    const int a = tid >> 4, b = tid & 15;
    // (a workgroup per FACE of a listed chunk: the six faces of a chunk one after the other were six chains of dependent loads and atomics —
    // 9 us of an edit's resolve for the five chunks a bite leaves with several regions)
    for (uint32_t item = bid; item < n_multi * 6u; item += nb) {
        const uint32_t chunk = multi_list[item / 6u];
        const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
        const uint8_t* own = labels + (size_t)chunk * IVX_CHUNK_VOXELS;  // a chunk with several regions is NonUniform: it has planes
        {
            const int f = (int)(item % 6u);
            const int dim = f >> 1, up = f & 1;
            const int ni = ci + (dim == 0 ? (up ? 1 : -1) : 0), nj = cj + (dim == 1 ? (up ? 1 : -1) : 0), nk = ck + (dim == 2 ? (up ? 1 : -1) : 0);
            if (ni < 0 || nj < 0 || nk < 0 || ni >= (int)g.cx || nj >= (int)g.cy || nk >= (int)g.cz) continue;
            const uint32_t nchunk = (ni * g.cy + nj) * g.cz + nk;
            const ivx_chunk_info ninfo = g.info[nchunk];
            if (ninfo.region_count == 0) continue;
            const uint8_t* nb = labels + (size_t)nchunk * IVX_CHUNK_VOXELS;
            const uint32_t so = up ? 15u : 0u, sn = up ? 0u : 15u;  // own / neighbour layer along `dim`
            const uint32_t oo = dim == 0 ? ((so << 8) | (a << 4) | b) : (dim == 1 ? ((a << 8) | (so << 4) | b) : ((a << 8) | (b << 4) | so));
            const uint32_t on = dim == 0 ? ((sn << 8) | (a << 4) | b) : (dim == 1 ? ((a << 8) | (sn << 4) | b) : ((a << 8) | (b << 4) | sn));
            const uint32_t la = own[oo];
            // a Uniform neighbour is region 0 everywhere and has no label plane (compact planes)
            const uint32_t lb = ninfo.kind == KIND_NONUNIFORM ? (uint32_t)nb[on] : 0u;
            const bool both = la != 255u && lb != 255u;
            const uint32_t pair = both ? ((la << 8) | lb) : 0xFFFFFFFFu;
            const uint32_t prev = __shfl_up(pair, 1, 64);
            const bool dup = (tid & 63u) != 0 && prev == pair;
            if (both && !dup) g_union(rparent, chunk * 256u + la, nchunk * 256u + lb);
        }
    }
}

// flatten the forest and count the roots per chunk (one thread per chunk: almost every chunk has 0-2 regions)
__device__ __forceinline__ void role_ccl_flatten(uint32_t bid, uint32_t nb, GridView g, uint32_t* __restrict__ rparent, uint32_t* __restrict__ root_counts,
                                                     uint32_t* __restrict__ group_sums) {
    __shared__ uint32_t s_w[4];
    const uint32_t chunk = bid * 256u + threadIdx.x;
    const bool live = chunk < g.cx * g.cy * g.cz;
    const uint32_t rc = live ? g.info[chunk].region_count : 0u;
    uint32_t n = 0;
    // A chunk with many regions (an edit's rough surface: a few hundred) is taken by its whole wave, a region per lane — one thread walking
    // 250 chains one after the other was 50 us of the edit's resolve, with the other 63 lanes of its wave done after two.
    constexpr uint32_t WIDE = 8u;
    for (uint32_t r = 0; r < (rc > WIDE ? 0u : rc); ++r) {
        const uint32_t node = chunk * 256u + r;
        const uint32_t root = g_find(rparent, node);
        if (root == node) n += 1;
        // safe while other threads still walk the forest: the parent only moves closer to the root
        else __hip_atomic_store(rparent + node, root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    {
        const uint32_t lane_ = threadIdx.x & 63u;
        for (unsigned long long wide = __ballot(rc > WIDE); wide; wide &= wide - 1ull) {  // (wave-uniform)
            const uint32_t owner = (uint32_t)__ffsll((long long)wide) - 1u;
            const uint32_t oc = (uint32_t)__shfl((int)chunk, (int)owner, 64), orc = (uint32_t)__shfl((int)rc, (int)owner, 64);
            uint32_t roots = 0;
            for (uint32_t r0 = 0; r0 < orc; r0 += 64u) {
                const uint32_t r = r0 + lane_;
                bool is_root = false;
                if (r < orc) {
                    const uint32_t node = oc * 256u + r;
                    const uint32_t root = g_find(rparent, node);
                    is_root = root == node;
                    if (!is_root) __hip_atomic_store(rparent + node, root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                roots += (uint32_t)__popcll(__ballot(is_root));
            }
            if (lane_ == owner) n = roots;
        }
    }
    // exclusive prefix of the root counts inside this group of 256 chunks (ordered) and the group's total: the two levels of
    // the scan; k_ccl_assign adds the totals of the groups before
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o) incl += t;
    }
    if (lane == 63u) s_w[wave] = incl;
    __syncthreads();
    const uint32_t w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    const uint32_t wbase = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
    if (live) root_counts[chunk] = wbase + incl - n;  // (the count itself is not needed again)
    if (threadIdx.x == 0) group_sums[bid] = (w0 + w1) + (w2 + w3);
}

// Grids of more than ASSIGN_MAX_GROUPS x 256 chunks: a launch of its own adds the totals of the groups before a chunk's group to
// the in-group prefix k_ccl_flatten left (every block adds up those totals itself).
__device__ __forceinline__ void role_scan_groups(uint32_t bid, uint32_t nb, uint32_t n, const uint32_t* __restrict__ in, const uint32_t* __restrict__ group_sums,
                                                     uint32_t* __restrict__ out, uint32_t* __restrict__ total) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t part = 0;
    for (uint32_t b = tid; b < bid; b += 256u) part += group_sums[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if (lane == 0) s_w[wave] = part;
    __syncthreads();
    if (tid == 0) s_base = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    __syncthreads();
    const uint32_t base = s_base;
    const uint32_t c = bid * 256u + tid;
    if (c < n) out[c] = base + in[c];
    if (bid == nb - 1 && tid == 0) *total = base + group_sums[bid];
}

// component ids: rank of the root node in (chunk, region) order; a non-root takes the id of its root, computed from the
// root's chunk offset and the root's rank among the roots of that chunk
#ifndef IVX_ASSIGN_MAX_GROUPS
#define IVX_ASSIGN_MAX_GROUPS 2048
#endif
constexpr uint32_t ASSIGN_MAX_GROUPS = IVX_ASSIGN_MAX_GROUPS;  // 524 288 chunks; larger grids take the k_scan_groups path
// FUSED: the scan over the group totals happens here (every block scans the few totals itself in LDS), root_offsets[c] is the
// prefix inside c's group; otherwise root_offsets[c] is the full prefix from k_scan_groups.
template <bool FUSED>
__device__ __forceinline__ void role_ccl_assign(uint32_t bid, uint32_t nb, GridView g, const uint32_t* __restrict__ rparent, const uint32_t* __restrict__ root_offsets,
                                                    const uint32_t* __restrict__ group_sums, uint32_t n_groups, uint32_t* __restrict__ rcompid,
                                                    uint32_t* __restrict__ total) {
    __shared__ uint32_t s_gpre[FUSED ? ASSIGN_MAX_GROUPS : 1];
    __shared__ uint32_t s_carry;
    if (FUSED) {
        // exclusive scan of the group totals, 256 at a time
        if (threadIdx.x == 0) s_carry = 0;
        __syncthreads();
        for (uint32_t g0 = 0; g0 < n_groups; g0 += 256u) {
            const uint32_t gi = g0 + threadIdx.x;
            const uint32_t v = gi < n_groups ? group_sums[gi] : 0u;
            uint32_t tot;
            __shared__ uint32_t s_ws[4];
            const uint32_t ex = prefix_ordered(v, s_ws, threadIdx.x, tot);
            if (gi < n_groups) s_gpre[gi] = s_carry + ex;
            __syncthreads();
            if (threadIdx.x == 0) s_carry += tot;
            __syncthreads();
        }
        if (bid == 0 && threadIdx.x == 0) *total = s_carry;
    }
    const uint32_t chunk = bid * 256u + threadIdx.x;
    if (chunk >= g.cx * g.cy * g.cz) return;
    const uint32_t rc = g.info[chunk].region_count;
    for (uint32_t r = 0; r < rc; ++r) {
        const uint32_t node = chunk * 256u + r;
        const uint32_t root = rparent[node];  // flattened: the root itself
        const uint32_t rchunk = root >> 8, rr = root & 255u;
        uint32_t rank = 0;
        for (uint32_t q = 0; q < rr; ++q) rank += rparent[rchunk * 256u + q] == rchunk * 256u + q;
        rcompid[node] = root_offsets[rchunk] + rank + (FUSED ? s_gpre[rchunk >> 8] : 0u);
    }
}

// The distinct (own component, neighbour's component) pairs across one x face. A workgroup takes FACE_COLS chunk columns — every thread the
// same face voxel of each, the columns' three dependent loads (chunk kind, label, component id) side by side — and collects the pairs of small
// ids (the usual case: a handful of components per slab) in a 64 x 64 bit table in LDS; one thread per table word then merges it into the
// slab's table with one atomic and lists the pairs whose bit it set first. (One workgroup per column and one atomic per wave on the slab's
// table was 16 us for a 32 x 32 face: a face inside one body is thousands of times the same pair, a thousand workgroups find the bit clear
// at the same moment, and that many atomics on one word queue.)
// `bid`: the workgroup's index among the role's (FACE_COLS columns each); `s_seen`: 128 words of LDS the caller lends.
constexpr uint32_t FACE_COLS = 8;
__device__ __forceinline__ void role_face_pairs(uint32_t bid, const GridView& g, uint32_t side, const uint8_t* __restrict__ labels,
                                                const uint32_t* __restrict__ rcompid, const uint16_t* __restrict__ nbr, uint32_t* __restrict__ n_pairs,
                                                uint2* __restrict__ pairs, uint32_t cap, uint32_t* __restrict__ seen, uint32_t* s_seen) {
    const uint32_t tid = threadIdx.x, cols = g.cy * g.cz;
    if (tid < 128u) s_seen[tid] = 0u;
    uint32_t col[FACE_COLS], chunk[FACE_COLS], kind[FACE_COLS], l[FACE_COLS], a[FACE_COLS], b[FACE_COLS];
#pragma unroll
    for (uint32_t c = 0; c < FACE_COLS; ++c) {
        col[c] = min(bid * FACE_COLS + c, cols - 1u);  // (a column past the end repeats the last one's loads and lists nothing)
        chunk[c] = (side ? g.cx - 1 : 0u) * cols + col[c];
        kind[c] = g.info[chunk[c]].kind;
    }
#pragma unroll
    for (uint32_t c = 0; c < FACE_COLS; ++c) {
        l[c] = labels[(size_t)chunk[c] * IVX_CHUNK_VOXELS + ((side ? 15u : 0u) << 8) + tid];  // (the plane is there whatever the kind; only its content may be stale)
        b[c] = nbr[(size_t)col[c] * 256 + tid];
    }
#pragma unroll
    for (uint32_t c = 0; c < FACE_COLS; ++c) {
        if (kind[c] != KIND_NONUNIFORM) l[c] = ivx_uniform_label(kind[c]);
        a[c] = rcompid[chunk[c] * 256u + l[c]];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t c = 0; c < FACE_COLS; ++c) {
        const uint32_t ca = l[c] == 255u ? NODE_NONE : a[c], cb = b[c] == 0xFFFFu ? NODE_NONE : b[c];
        const bool both = ca != NODE_NONE && cb != NODE_NONE && bid * FACE_COLS + c < cols;
        // drop repeats along the lane order (one wave = four rows of 16 face voxels)
        const uint32_t pa = __shfl_up(ca, 1, 64), pb = __shfl_up(cb, 1, 64);
        const bool dup = (tid & 63u) != 0 && pa == ca && pb == cb;
        if (both && !dup) {
            if (seen && ca < 64u && cb < 64u) {
                const uint32_t bit = ca * 64u + cb;
                if (!((__hip_atomic_load(&s_seen[bit >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (bit & 31u)) & 1u)) atomicOr(&s_seen[bit >> 5], 1u << (bit & 31u));
            } else {
                const uint32_t slot = atomicAdd(n_pairs, 1u);
                if (slot < cap) pairs[slot] = make_uint2(ca, cb);
            }
        }
    }
    __syncthreads();
    if (tid < 128u && seen) {
        const uint32_t mine = s_seen[tid];
        if (mine) {
            uint32_t fresh = mine & ~__hip_atomic_load(&seen[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a look before the atomic)
            if (fresh) fresh &= ~atomicOr(&seen[tid], fresh);
            while (fresh) {
                const uint32_t bit = tid * 32u + (uint32_t)(__ffs(fresh) - 1);
                fresh &= fresh - 1u;
                const uint32_t slot = atomicAdd(n_pairs, 1u);
                if (slot < cap) pairs[slot] = make_uint2(bit >> 6, bit & 63u);
            }
        }
    }
}

// Everything the other ranks need from this slab after a step, as one fixed-size record of 64-bit words written on the
// device so that it can go straight into an all-gather (impact_amd/distributed.py): [0] components, [1] pairs across the
// upper face, [2..14) occupied ranges (public layout, global coordinates), [14..17) mesh totals, [18..28) moments (f64
// bit patterns), [28..28+2*max_pairs) the pairs.
__device__ __forceinline__ void role_step_record(const uint32_t* __restrict__ rscalar, const uint32_t* __restrict__ pair_count,
                                                 const uint2* __restrict__ pairs, const uint32_t* __restrict__ mesh_totals,
                                                 const double* __restrict__ moments, uint32_t x_off, uint32_t max_pairs,
                                                 unsigned long long* __restrict__ rec) {
    const uint32_t tid = threadIdx.x;
    const uint32_t np = pair_count ? min(pair_count[0], 0xFFFFFFFFu) : 0u;
    if (tid == 0) {
        rec[0] = rscalar[0];
        rec[1] = np;  // may exceed max_pairs: the reader reports the overflow
        const uint32_t* raw = rscalar + 16;
        if (raw[6] == 0) {
            for (int i = 0; i < 12; ++i) rec[2 + i] = 0;
        } else {
            for (int d = 0; d < 3; ++d) {
                rec[2 + 2 * d] = raw[d] + (d == 0 ? x_off : 0u);
                rec[3 + 2 * d] = raw[6 + d] + (d == 0 ? x_off : 0u);
                rec[8 + 2 * d] = raw[3 + d] + (d == 0 ? x_off * 16u : 0u);
                rec[9 + 2 * d] = raw[9 + d] + (d == 0 ? x_off * 16u : 0u);
            }
        }
        rec[14] = mesh_totals[0];
        rec[15] = mesh_totals[1];
        rec[16] = mesh_totals[2];
        rec[17] = rscalar[1];  // error flags
    }
    if (tid < 10) rec[18 + tid] = (unsigned long long)__double_as_longlong(moments[tid]);
    for (uint32_t i = tid; i < min(np, max_pairs); i += 256u) {
        rec[28 + 2 * i] = pairs[i].x;
        rec[29 + 2 * i] = pairs[i].y;
    }
}

}  // namespace ivx_roles
