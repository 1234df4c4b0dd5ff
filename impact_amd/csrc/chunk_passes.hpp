// Per-chunk passes shared by the stand-alone kernels (ccl.hip, inertia.hip) and the fused sweep in derive.hip: the
// chunk-local connected regions and the chunk's moments need nothing but the chunk's own non-empty row masks (and types),
// which k_derive holds in registers anyway.
#pragma once
// developer build (make EXTRA=-DIVX_DERIVE_DEBUG, tools/derive_budget.sh): store groups of the derive sweep switched off one by one at run time
// (IVX_DERIVE_SKIP, a bit per group) — what each of them costs in counter traffic. Production builds compile the test away.
#ifdef IVX_DERIVE_DEBUG
static __device__ uint32_t ivx_dbg_derive_skip = 0u;
#define IVX_DBG_KEEP(bit) (!(ivx_dbg_derive_skip & (bit)))
#else
#define IVX_DBG_KEEP(bit) true
#endif
#include "ivx_internal.hpp"

#define NODE_NONE 0xFFFFFFFFu

// (path halving on the way: the larger root always goes under the smaller, so every parent is an ancestor with a smaller index and a
// parent may be replaced by any of its own ancestors at any time; the roots, all that is counted afterwards, are untouched)
// (relaxed workgroup-scope atomics, not `volatile`: every access is made, none is reordered against the union's atomicMin on the same word,
// and — unlike a volatile access, which the compiler brackets with a wait for EVERY outstanding memory operation — a read waits for LDS
// only: k_derive has the next chunk's global loads in flight across this pass)
__device__ __forceinline__ uint32_t lds_find(uint32_t* par, uint32_t x) {
    uint32_t p;
    while ((p = __hip_atomic_load(par + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != x) {
        const uint32_t gp = __hip_atomic_load(par + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (gp != p) __hip_atomic_store(par + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        x = gp;
    }
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
    // small words of the exact numbering (ccl_exact_chunk)
    uint32_t bm[44];  // bitmap over the 1352 positions of the boundary walk
    uint32_t w[8];
    uint32_t misc[4];
};

// ---- exact chunk-local numbering for chunks with several regions -------------------------------------------------------------------
// The reference numbers the regions of a chunk in an order that depends on which voxel its SEQUENTIAL union-find leaves as the root
// of each set (split_detection.rs:700-831): voxels in (i,j,k) order, and for every voxel v the roots of its +x, +y, +z neighbours are
// attached under the root of v's set (assign_parent, 1776-1782) — "the set of the voxel being processed wins". A set that touches
// the chunk boundary is numbered when the six face loops (Loop3::over_full_boundary, utils.rs:247-322) reach its ROOT voxel if the
// root lies on the boundary, else at the first face voxel of the set they visit (761-801); sets that do not touch the boundary follow
// in (i,j,k) order of their roots (815-831). Raw label values are compared bit for bit with the reference's, so the roots must be the
// sequential ones. They are, without replaying the sequence voxel by voxel:
//   * a voxel with a non-empty lower neighbour (-x, -y or -z) belongs, when its turn comes, to the set its FIRST-processed lower
//     neighbour put it in (fp(v) = v-256, else v-16, else v-1); a voxel with none is a singleton then: a SOURCE. Every root is a source.
//     src0(v) = the source reached along fp: pointer jumping, all voxels at once;
//   * sets merge only when a voxel a meets an upper neighbour b that an earlier lower neighbour already put elsewhere
//     (fp(b) != a): then a's set wins. These EVENTS, in the order of a, are a sequential union-find over the few sources; they are
//     listed by ordered compaction (cross-basin ones only) and applied by one wave, 64 at a time: every lane looks up its two sets,
//     the earliest event between different sets is applied, the rest look again (one round per effective merge);
//   * numbering: a key per set (walk position of the boundary root, else the smallest walk position of its boundary voxels), ranks
//     from a bitmap over the 1352 walk positions; interior-only sets by rank of their root voxel.
// All of it reuses the 16 KB of `sh.par`: ptr u16[4096] | spar u16[2048] (sources are never adjacent along k: one slot per voxel
// pair) | events 2 x u16[768] | key u32[256].
__device__ __forceinline__ uint32_t ivx_walk_pos(uint32_t idx) {  // position of a boundary voxel in Loop3::over_full_boundary; 0xFFFF inside
    const uint32_t i = idx >> 8, j = (idx >> 4) & 15u, k = idx & 15u;
    if (i == 0u) return idx & 255u;
    if (i == 15u) return 256u + (idx & 255u);
    if (j == 0u) return 512u + (i - 1u) * 16u + k;
    if (j == 15u) return 736u + (i - 1u) * 16u + k;
    if (k == 0u) return 960u + (i - 1u) * 14u + (j - 1u);
    if (k == 15u) return 1156u + (i - 1u) * 14u + (j - 1u);
    return 0xFFFFu;
}

// developer probes of the exact numbering (make TRACE=1; read back by ivx_debug_exact_trace of the translation unit whose kernel ran it)
#ifdef IVX_WG_TRACE
static __device__ unsigned long long ivx_exact_trace_buf[64 * 8];
#define IVX_XT(slot)                                                                                  \
    do {                                                                                              \
        if (threadIdx.x == 0) ivx_exact_trace_buf[(chunk & 63u) * 8 + (slot)] = wall_clock64();       \
    } while (0)
#else
#define IVX_XT(slot) \
    do {             \
    } while (0)
#endif

// ---- helpers of ccl_exact_chunk: a thread's row of sixteen u16 entries as eight packed words ------------------------------------------
__device__ __forceinline__ uint32_t ivx_h16(const uint32_t* w, int k) { return (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xFFFFu); }
__device__ __forceinline__ void ivx_row16_load(const uint16_t* row, uint32_t* w) {
    const uint4 lo = reinterpret_cast<const uint4*>(row)[0], hi = reinterpret_cast<const uint4*>(row)[1];
    w[0] = lo.x, w[1] = lo.y, w[2] = lo.z, w[3] = lo.w, w[4] = hi.x, w[5] = hi.y, w[6] = hi.z, w[7] = hi.w;
}
__device__ __forceinline__ void ivx_row16_store(uint16_t* row, const uint32_t* w) {
    reinterpret_cast<uint4*>(row)[0] = make_uint4(w[0], w[1], w[2], w[3]);
    reinterpret_cast<uint4*>(row)[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// out[k] = table[in[k] >> shift] for the sixteen entries (independent loads: one LDS latency for the row instead of sixteen)
__device__ __forceinline__ void ivx_row16_gather(const uint16_t* table, const uint32_t* in, uint32_t shift, uint32_t* out) {
    uint32_t t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = table[ivx_h16(in, k) >> shift];
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = t[2 * i] | (t[2 * i + 1] << 16);
}

// The merge events of a chunk with at most 64 K sources, applied in order by ONE wave with the sets in registers: lane l of R[q] holds the
// set (the index of its root source) of source 64 q + l. An event (a, b) between different sets makes a's root the root of the union
// (assign_parent, split_detection.rs:1776-1782): every source of b's set takes a's root — one compare-and-select over the lanes, no chain to
// follow, nothing in memory. Per round every lane looks up its event's two sets (a lane permute each); the EARLIEST event between different
// sets is the next the sequential order would apply (the ones before it join equal sets: nothing to do), so it is applied and the lanes
// follow the merge in their own registers: one round per effective merge, a few dozen cycles each. (The union-find in LDS this replaces — finds with path halving, a ballot
// and a conflict scan among the pending lanes per round — took 63 us for a chunk of 65 sources and 796 events; this takes ~5.)
template <int K>
__device__ __forceinline__ uint32_t ivx_exact_set_of(const uint32_t* R, uint32_t s) {
    uint32_t v = (uint32_t)__shfl((int)R[0], (int)(s & 63u), 64);
#pragma unroll
    for (int q = 1; q < K; ++q) {
        const uint32_t t = (uint32_t)__shfl((int)R[q], (int)(s & 63u), 64);
        v = (s >> 6) == (uint32_t)q ? t : v;
    }
    return v;
}
template <int K>
__device__ __forceinline__ void ivx_exact_apply_events(uint32_t* R, const uint16_t* ev_a, const uint16_t* ev_b, uint32_t n_win, uint32_t lane) {
    for (uint32_t e0 = 0; e0 < n_win; e0 += 64u) {
        const bool valid = e0 + lane < n_win;
        const uint32_t sa = valid ? ev_a[e0 + lane] : 0u, sb = valid ? ev_b[e0 + lane] : 0u;
        // (the lanes look their events' sets up once per batch — a lane permute per register of R — and from then on follow the merges
        // themselves: whoever holds the loser holds the winner afterwards)
        uint32_t ra = ivx_exact_set_of<K>(R, sa), rb = ivx_exact_set_of<K>(R, sb);
        for (int guard = 0; guard < 64 * K + 2; ++guard) {  // (a merge takes a set away: at most 64 K - 1 of them in all)
            const unsigned long long pending = __ballot(valid && ra != rb);
            if (!pending) break;
            const int j = __ffsll((long long)pending) - 1;  // (wave-uniform)
            const uint32_t win = (uint32_t)__builtin_amdgcn_readlane((int)ra, j), lose = (uint32_t)__builtin_amdgcn_readlane((int)rb, j);
            ra = ra == lose ? win : ra;
            rb = rb == lose ? win : rb;
#pragma unroll
            for (int q = 0; q < K; ++q) R[q] = R[q] == lose ? win : R[q];
        }
    }
}

// All 256 threads; sh.mask[] holds the non-empty masks of the 256 rows (thread = row (i,j), bit = k). Writes the label plane, the
// chunk's region table and returns (region_count, boundary_region_count), both saturated at 254 with error bit 1 set beyond.
// A thread keeps its row's sixteen pointers in eight packed registers and follows them by GATHERS — sixteen independent LDS loads, one
// latency — wherever the straightforward loop over the set bits of its mask would chain two to four dependent loads per voxel (the kernel
// that hosts this is held to 64 VGPRs: the rows stay packed).
__device__ __forceinline__ void ccl_exact_chunk(CclShared& sh, uint32_t tid, uint32_t chunk, uint8_t* __restrict__ labels, uint32_t* __restrict__ rparent,
                                                uint32_t* __restrict__ rscalar, uint32_t& rc_out, uint32_t& brc_out) {
    constexpr uint32_t EV_CAP = 768u;
    uint16_t* ptr = reinterpret_cast<uint16_t*>(sh.par);          // [4096]
    uint16_t* spar = ptr + IVX_CHUNK_VOXELS;                      // [2048], slot v >> 1
    uint16_t* ev_a = spar + 2048;                                 // [768]
    uint16_t* ev_b = ev_a + EV_CAP;                               // [768]
    uint32_t* key = reinterpret_cast<uint32_t*>(ev_b + EV_CAP);   // [256]
    uint16_t* src_voxel = reinterpret_cast<uint16_t*>(sh.mask);   // [256] voxel of source number s (once every thread has its masks)
    const uint32_t ti = tid >> 4, tj = tid & 15u, lane = tid & 63u, wave = tid >> 6;
    const uint32_t m = sh.mask[tid];
    const uint32_t mx = ti > 0u ? sh.mask[tid - 16u] : 0u;  // row below in x, in y
    const uint32_t my = tj > 0u ? sh.mask[tid - 1u] : 0u;
    const uint32_t m_yu = tj < 15u ? sh.mask[tid + 1u] : 0u;                   // row above in y
    const uint32_t m_xd_yu = (ti > 0u && tj < 15u) ? sh.mask[tid - 15u] : 0u;  // ... and the row below that one in x
    const uint32_t base_v = tid * 16u;
    __syncthreads();  // everybody has read what it needs from the union-find of ccl_local_chunk and its masks: sh.par and sh.mask are free
    IVX_XT(0);
    // fp pointers; sources point to themselves (empty voxels too: never followed)
    const uint32_t has_xd = m & mx, has_yd = m & my & ~mx, has_zd = m & (m << 1) & ~mx & ~my;
    const uint32_t sources = m & ~mx & ~my & ~(m << 1);
    uint32_t pp[8];  // the row's pointers, packed
    {
        uint32_t p[16];
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) {
            const uint32_t v = base_v + k;
            p[k] = ((has_xd >> k) & 1u) ? v - 256u : (((has_yd >> k) & 1u) ? v - 16u : (((has_zd >> k) & 1u) ? v - 1u : v));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) pp[i] = p[2 * i] | (p[2 * i + 1] << 16);
        ivx_row16_store(ptr + base_v, pp);
    }
    // the sources, numbered in voxel order
    uint32_t src_first, n_sources;
    {
        const uint32_t n_s = __popc(sources);
        uint32_t incl = n_s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= (uint32_t)o) incl += t;
        }
        if (lane == 63u) sh.w[wave] = incl;
        __syncthreads();
        const uint32_t w0 = sh.w[0], w1 = sh.w[1], w2 = sh.w[2], w3 = sh.w[3];
        src_first = (wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2))) + incl - n_s;
        n_sources = (w0 + w1) + (w2 + w3);
    }
    // up to 256 sources: the sets live in one wave's registers while the events are applied (spar: the source's number); beyond: the
    // union-find over spar (a source's parent source)
    const bool fast = n_sources <= 256u;
    {
        uint32_t s = sources, idx = src_first;
        while (s) {
            const uint32_t k = (uint32_t)__ffs(s) - 1u;
            s &= s - 1u;
            spar[(base_v + k) >> 1] = (uint16_t)(fast ? idx : base_v + k);
            if (fast) src_voxel[idx] = (uint16_t)(base_v + k);
            idx += 1u;
        }
    }
    if (tid < 44u) sh.bm[tid] = 0u;
    key[tid] = 0xFFFFFFFFu;
    __syncthreads();
    // pointer jumping (in place: a pointer only ever moves to an ancestor; a thread's own entries change by its own stores only, so its
    // registers stay current): chains are at most 45 long
    for (int round = 0; round < 6; ++round) {
        ivx_row16_gather(ptr, pp, 0u, pp);
        ivx_row16_store(ptr + base_v, pp);
        __syncthreads();
    }
    IVX_XT(1);
    // events of this row, in order: (a, b = a + 16) when b also has a -x neighbour; (a, b = a + 1) when b has a -x or -y neighbour
    const uint32_t ev_y_geom = m & m_yu & m_xd_yu;               // b = a + 16 has b - 256
    const uint32_t ev_z_geom = m & (m >> 1) & ((mx | my) >> 1);  // b = a + 1 has b - 256 or b - 16
    uint32_t evy = 0, evz = 0;
    uint32_t qq[8];  // the pointers of the row above in y
    ivx_row16_load(ptr + (tj < 15u ? base_v + 16u : base_v), qq);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        // (an event between the same two basins as the row's event just before it joins what that one joined: dropped — along the contact
        // of two basins every voxel pair would list one)
        if (((ev_y_geom >> k) & 1u) && ivx_h16(pp, k) != ivx_h16(qq, k) &&
            !(k > 0 && ((evy >> (k > 0 ? k - 1 : 0)) & 1u) && ivx_h16(pp, k) == ivx_h16(pp, k > 0 ? k - 1 : 0) && ivx_h16(qq, k) == ivx_h16(qq, k > 0 ? k - 1 : 0)))
            evy |= 1u << k;
        if (k < 15 && ((ev_z_geom >> k) & 1u) && ivx_h16(pp, k) != ivx_h16(pp, k + 1)) evz |= 1u << k;
    }
    uint32_t n_mine = __popc(evy) + __popc(evz), first, total;
    {   // ordered block prefix (thread order = voxel order)
        uint32_t incl = n_mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= (uint32_t)o) incl += t;
        }
        if (lane == 63u) sh.w[4 + wave] = incl;
        __syncthreads();
        const uint32_t w0 = sh.w[4], w1 = sh.w[5], w2 = sh.w[6], w3 = sh.w[7];
        first = (wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2))) + incl - n_mine;
        total = (w0 + w1) + (w2 + w3);
    }
    IVX_XT(2);
    if (total) {  // (workgroup-uniform)
        // what an event names: the sets' source numbers (fast) or source voxels
        uint32_t ea[8], eb[8];
        if (fast) {
            ivx_row16_gather(spar, pp, 1u, ea);
            ivx_row16_gather(spar, qq, 1u, eb);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) ea[i] = pp[i], eb[i] = qq[i];
        }
        uint32_t R[4] = {lane, 64u + lane, 128u + lane, 192u + lane};  // (wave 0's: the sets, see ivx_exact_apply_events)
        for (uint32_t win = 0; win < total; win += EV_CAP) {
            __syncthreads();  // the previous window is consumed
            {
                uint32_t slot = first;  // global index of this thread's next event
#pragma unroll
                for (int k = 0; k < 16; ++k) {  // per voxel: y before z (the reference's order inside a voxel; either way the same winner)
                    if ((evy >> k) & 1u) {
                        if (slot >= win && slot < win + EV_CAP) {
                            ev_a[slot - win] = (uint16_t)ivx_h16(ea, k);
                            ev_b[slot - win] = (uint16_t)ivx_h16(eb, k);
                        }
                        slot += 1u;
                    }
                    if (k < 15 && ((evz >> k) & 1u)) {
                        if (slot >= win && slot < win + EV_CAP) {
                            ev_a[slot - win] = (uint16_t)ivx_h16(ea, k);
                            ev_b[slot - win] = (uint16_t)ivx_h16(ea, k + 1);
                        }
                        slot += 1u;
                    }
                }
            }
            __syncthreads();
            if (wave == 0u) {
                const uint32_t n_win = min(EV_CAP, total - win);
                if (fast) {
                    if (n_sources <= 64u) ivx_exact_apply_events<1>(R, ev_a, ev_b, n_win, lane);
                    else ivx_exact_apply_events<4>(R, ev_a, ev_b, n_win, lane);
                } else {
                    for (uint32_t e0 = 0; e0 < n_win; e0 += 64u) {
                        const bool valid = e0 + lane < n_win;
                        const uint32_t pa = valid ? ev_a[e0 + lane] : 0u, pb = valid ? ev_b[e0 + lane] : 0u;
                        for (int guard = 0; guard < 4096; ++guard) {
                            uint32_t ra = pa, rb = pb;
                            if (valid) {
                                // (path halving: a set that keeps winning grows a chain as long as the number of sets it swallowed; a parent only
                                // ever moves to an ancestor, so the roots — all that the numbering depends on — are untouched)
                                for (uint32_t p; (p = spar[ra >> 1]) != ra;) {
                                    const uint32_t gp = spar[p >> 1];
                                    spar[ra >> 1] = (uint16_t)gp;
                                    ra = gp;
                                }
                                for (uint32_t p; (p = spar[rb >> 1]) != rb;) {
                                    const uint32_t gp = spar[p >> 1];
                                    spar[rb >> 1] = (uint16_t)gp;
                                    rb = gp;
                                }
                            }
                            const bool mine = valid && ra != rb;
                            const unsigned long long pending = __ballot(mine);
                            if (!pending) break;
                            // In event order a's set wins: the root of b's set becomes a LOSER (it gets a parent), and which root a merged set ends up
                            // with is settled by which of its roots never loses. Several pending events are applied in one round when that cannot
                            // change any event's loser: an event waits only for an earlier pending event whose loser is its loser or its winner, or
                            // whose winner is its loser (the first pending event never waits). Events that share a WINNER — the usual case, one
                            // big set swallowing many small ones — all go at once. Losers of a round are distinct and none is a winner of the
                            // round, so the forest stays a forest; the sets and their roots are those of the one-at-a-time order.
                            bool blocked = false;
                            for (unsigned long long pm = pending; pm; pm &= pm - 1ull) {
                                const uint32_t j = (uint32_t)__ffsll((long long)pm) - 1u;  // (wave-uniform)
                                const uint32_t raj = (uint32_t)__builtin_amdgcn_readlane((int)ra, (int)j), rbj = (uint32_t)__builtin_amdgcn_readlane((int)rb, (int)j);
                                blocked = blocked || (j < lane && (rbj == rb || rbj == ra || raj == rb));
                            }
                            if (mine && !blocked) spar[rb >> 1] = (uint16_t)ra;
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        }
                    }
                }
            }
        }
        if (fast && wave == 0u) {  // a source's parent: the root source of its set, directly
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t s = 64u * q + lane;
                if (s < n_sources) spar[src_voxel[s] >> 1] = src_voxel[R[q]];
            }
        }
    } else if (fast) {  // no events: every source is its own root
        uint32_t s = sources;
        while (s) {
            const uint32_t k = (uint32_t)__ffs(s) - 1u;
            s &= s - 1u;
            spar[(base_v + k) >> 1] = (uint16_t)(base_v + k);
        }
    }
    __syncthreads();
    IVX_XT(3);
#ifdef IVX_WG_TRACE
    if (threadIdx.x == 0) ivx_exact_trace_buf[(chunk & 63u) * 8 + 7] = total | (n_sources << 16);
#endif
    // flatten: ptr[source] = root source, so that root(v) = ptr[ptr[v]] for every non-empty voxel
    {
        uint32_t s = sources;
        while (s) {
            const uint32_t k = (uint32_t)__ffs(s) - 1u;
            s &= s - 1u;
            uint32_t r = base_v + k;
            for (uint32_t p; (p = spar[r >> 1]) != r;) r = p;
            ptr[base_v + k] = (uint16_t)r;
        }
    }
    __syncthreads();
    // dense ids of the roots in voxel order (spar is free now: id of root r at spar[r >> 1])
    uint32_t roots = 0;
    {
        uint32_t s = sources;
        while (s) {
            const uint32_t k = (uint32_t)__ffs(s) - 1u;
            s &= s - 1u;
            if (ptr[base_v + k] == base_v + k) roots |= 1u << k;
        }
    }
    uint32_t id0, n_sets;
    {
        const uint32_t n_r = __popc(roots);
        uint32_t incl = n_r;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= (uint32_t)o) incl += t;
        }
        __syncthreads();
        if (lane == 63u) sh.w[wave] = incl;
        __syncthreads();
        const uint32_t w0 = sh.w[0], w1 = sh.w[1], w2 = sh.w[2], w3 = sh.w[3];
        id0 = (wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2))) + incl - n_r;
        n_sets = (w0 + w1) + (w2 + w3);
        uint32_t r = roots, id = id0;
        while (r) {
            const uint32_t k = (uint32_t)__ffs(r) - 1u;
            r &= r - 1u;
            spar[(base_v + k) >> 1] = (uint16_t)min(id, 255u);  // (more than 254 sets is an error; ids stay inside the tables)
            id += 1u;
        }
    }
    __syncthreads();
    IVX_XT(4);
    // the root and the set id of each of the row's voxels (pp: a voxel's source, or the voxel itself when it is one — or empty, which nobody asks)
    uint32_t rr[8], idw[8];
    ivx_row16_gather(ptr, pp, 0u, rr);
    ivx_row16_gather(spar, rr, 1u, idw);
    // keys: the boundary voxels of this row
    {
        const bool edge_row = ti == 0u || ti == 15u || tj == 0u || tj == 15u;
        const uint32_t bnd = m & (edge_row ? 0xFFFFu : 0x8001u);
        uint32_t prev_id = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (!edge_row && k != 0 && k != 15) continue;  // (compile-time k: interior rows have their two end voxels only)
            if ((bnd >> k) & 1u) {
                const uint32_t v = base_v + (uint32_t)k;
                const uint32_t r = ivx_h16(rr, k), id = ivx_h16(idw, k) & 255u;
                const bool root_on_boundary = ivx_walk_pos(r) != 0xFFFFu;
                if (root_on_boundary) {
                    if (r == v) key[id] = ivx_walk_pos(v);  // numbered when the walk reaches the root itself
                } else if (id != prev_id) {  // (walk positions grow with k inside a row: the first voxel of a stretch of one set has the smallest)
                    atomicMin(&key[id], ivx_walk_pos(v));
                }
                prev_id = id;
            }
        }
    }
    __syncthreads();
    // numbers: thread = set id
    const uint32_t my_key = tid < n_sets ? key[tid] : 0xFFFFFFFFu;
    const bool touches = my_key != 0xFFFFFFFFu;
    if (touches) atomicOr(&sh.bm[my_key >> 5], 1u << (my_key & 31u));
    uint32_t inner_rank, n_inner;
    {   // rank of the interior-only sets among themselves (ids are in root order)
        const uint32_t is_inner = (tid < n_sets && !touches) ? 1u : 0u;
        uint32_t incl = is_inner;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= (uint32_t)o) incl += t;
        }
        __syncthreads();
        if (lane == 63u) sh.w[wave] = incl;
        __syncthreads();
        const uint32_t w0 = sh.w[0], w1 = sh.w[1], w2 = sh.w[2], w3 = sh.w[3];
        inner_rank = (wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2))) + incl - is_inner;
        n_inner = (w0 + w1) + (w2 + w3);
    }
    uint32_t n_boundary = 0;
    {
        uint32_t below = 0;
#pragma unroll
        for (uint32_t wd = 0; wd < 43u; ++wd) {
            const uint32_t bits = sh.bm[wd];
            n_boundary += __popc(bits);
            if (touches && wd < (my_key >> 5)) below += __popc(bits);
            else if (touches && wd == (my_key >> 5)) below += __popc(bits & ((1u << (my_key & 31u)) - 1u));
        }
        // label of the set: saturating like the reference's counter (min(current + 1, 255))
        const uint32_t number = touches ? below : n_boundary + inner_rank;
        __syncthreads();  // every thread has read its key: the table now holds the numbers
        if (tid < n_sets) key[tid] = min(number, 255u);
    }
    (void)n_inner;
    __syncthreads();
    {
        uint32_t lab[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) lab[k] = key[ivx_h16(idw, k) & 255u];
        uint32_t w4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w4[q] = 0u;
#pragma unroll
            for (int b = 0; b < 4; ++b) w4[q] |= (((m >> (4 * q + b)) & 1u) ? (lab[4 * q + b] & 0xFFu) : 0xFFu) << (8 * b);
        }
        *reinterpret_cast<uint4*>(labels + (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)tid * 16) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
    uint32_t total_sets = n_sets;
    if (total_sets > 254u) {
        if (tid == 0) atomicOr(&rscalar[1], 1u);
        total_sets = 254u;
    }
    rparent[(size_t)chunk * 256 + tid] = tid < total_sets ? chunk * 256u + tid : NODE_NONE;
    rc_out = total_sets;
    brc_out = n_boundary < 254u ? n_boundary : 254u;
    IVX_XT(5);
    __syncthreads();  // sh is reused by the caller's next chunk
}

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
    // One barrier publishes three things: the workgroup's three votes (a ballot per wave, four words of LDS — the previous user of these
    // words is at least one barrier back): every row full? any voxel at all? any voxel on the chunk's boundary?; the rows' masks; and
    // the union-find's nodes — one per run, keyed by the voxel index of its first voxel (a chunk that the votes settle wrote its few
    // nodes for nothing).
    const bool edge_row = ti == 0 || ti == 15 || tj == 0 || tj == 15;
    const uint32_t starts = m & ~(m << 1);
    int all_full = kind != KIND_VOID, any = all_full, touches = 0;
    sh.mask[tid] = m;
    {
        uint32_t r = starts;  // (none for a chunk that is known)
        while (r) {
            const int k = __ffs(r) - 1;
            r &= r - 1;
            sh.par[tid * 16 + k] = tid * 16 + k;
        }
    }
    if (tid == 0) sh.cnt = 0;
    bool slab = false;  // one region by the layer test below
    if (!known) {
        // (with the votes: does a row hold several runs? which layers k does some non-empty row lack? which rows are non-empty?)
        const unsigned long long occ_b = __ballot(m != 0u);
        const uint32_t mine = (__ballot(m != 0xFFFFu) ? 1u : 0u) | (occ_b ? 2u : 0u) |
                              (__ballot(edge_row ? (m != 0u) : ((m & 0x8001u) != 0u)) ? 4u : 0u) | (__ballot((starts & (starts - 1u)) != 0u) ? 8u : 0u);
        uint32_t lack = m ? (~m & 0xFFFFu) : 0u;
        lack |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lack, 0x111, 0xF, 0xF, true);  // row_shr:1 .. 8: lane 15 of a DPP row has the row's OR
        lack |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lack, 0x112, 0xF, 0xF, true);
        lack |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lack, 0x114, 0xF, 0xF, true);
        lack |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lack, 0x118, 0xF, 0xF, true);
        const uint32_t wave_lack = ((uint32_t)__builtin_amdgcn_readlane((int)lack, 15) | (uint32_t)__builtin_amdgcn_readlane((int)lack, 31)) |
                                   ((uint32_t)__builtin_amdgcn_readlane((int)lack, 47) | (uint32_t)__builtin_amdgcn_readlane((int)lack, 63));
        if ((tid & 63u) == 0u) {
            sh.w[tid >> 6] = mine;
            sh.w[4 + (tid >> 6)] = wave_lack;
            sh.bm[2 * (tid >> 6)] = (uint32_t)occ_b;  // (the walk bitmap of the exact numbering: not in use in this pass)
            sh.bm[2 * (tid >> 6) + 1] = (uint32_t)(occ_b >> 32);
        }
        __syncthreads();
        const uint32_t votes = (sh.w[0] | sh.w[1]) | (sh.w[2] | sh.w[3]);
        all_full = !(votes & 1u);
        any = (votes & 2u) != 0u;
        touches = (votes & 4u) != 0u;
        // One region without the union-find: every non-empty row is ONE run, some layer k is in all of them — so two non-empty rows next to
        // each other always touch, and the region is connected when the SET of non-empty rows is, as a figure in the (i, j) plane — and that
        // set is one run of j per i, overlapping from each i to the next, over one run of i (sufficient, not necessary: anything else takes
        // the general path). A slab of material crossing the chunk — a floor, a wall's face, a plate — is this case; the union-find with its
        // two barriers was a third of such a chunk's time in the sweep.
        const uint32_t lack_all = (sh.w[4] | sh.w[5]) | (sh.w[6] | sh.w[7]);
        if (any && !all_full && !(votes & 8u) && (~lack_all & 0xFFFFu) != 0u) {  // (workgroup-uniform)
            const uint32_t i = tid & 15u, i1 = i < 15u ? i + 1u : 15u;
            const uint32_t b = (sh.bm[i >> 1] >> (16u * (i & 1u))) & 0xFFFFu;                            // non-empty rows j of line i
            const uint32_t bn = i < 15u ? ((sh.bm[i1 >> 1] >> (16u * (i1 & 1u))) & 0xFFFFu) : 0u;       // ... of line i + 1
            const uint32_t runs = b & ~(b << 1);
            const bool ok = b == 0u || ((runs & (runs - 1u)) == 0u && (bn == 0u || (b & bn) != 0u));
            const uint32_t lines = (uint32_t)__ballot(b != 0u) & 0xFFFFu, line_runs = lines & ~(lines << 1);
            slab = __ballot(!ok) == 0ull && (line_runs & (line_runs - 1u)) == 0u;
        }
    }
    if (!any || all_full) {
        // no voxels, or one solid region touching every face
        const uint32_t lab = any ? 0u : 0xFFFFFFFFu;
        if (kind == KIND_NONUNIFORM && IVX_DBG_KEEP(8u)) *reinterpret_cast<uint4*>(labels + base) = make_uint4(lab, lab, lab, lab);  // else: compact planes
        // only slots below region_count are ever read (flatten / assign / find walk valid nodes only)
        if (tid == 0) rp[0] = any ? chunk * 256u : NODE_NONE;
        rc_out = any ? 1u : 0u;
        brc_out = any ? 1u : 0u;
        return;
    }
    // 2. join runs across +x and +y: one union per run of the overlap between the two rows
    if (!slab) {
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
    uint32_t rc = 1u;
    if (!slab) {  // (workgroup-uniform)
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
    __syncthreads();
    rc = sh.cnt;
    }
    if (rc == 1u) {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (!((m >> k) & 1u)) w[k >> 2] |= 0xFFu << (8 * (k & 3));
        if (IVX_DBG_KEEP(8u)) *reinterpret_cast<uint4*>(labels + base) = make_uint4(w[0], w[1], w[2], w[3]);
        if (tid == 0) rp[0] = chunk * 256u;
        rc_out = 1u;
        brc_out = touches ? 1u : 0u;
    } else {
        // several regions: the reference's numbering is reproduced by role_ccl_local_exact right after this sweep (it also sets the
        // boundary count); the list also names the chunks whose regions k_ccl_merge_multi joins with all six neighbours
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
// inclusive sums along every DPP row of 16 lanes; lane 15 of a row holds the row's total
__device__ __forceinline__ double ivx_row16_sum_f64(double v) {
    v += ivx_dpp_f64<0x111>(v);  // row_shr:1
    v += ivx_dpp_f64<0x112>(v);  // row_shr:2
    v += ivx_dpp_f64<0x114>(v);  // row_shr:4
    v += ivx_dpp_f64<0x118>(v);  // row_shr:8
    return v;
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
// k0 the chunk's first k; s_dens the 256 densities and s_red[16][10] scratch in LDS. Ends with the ten sums in `out10`.
// the row's three sums over its non-empty voxels k: density, density * (2K + 1), density * (3K^2 + 3K + 1), K = k0 + k
__device__ __forceinline__ void moments_row_sums(uint32_t m, const uint32_t tw[4], const float* s_dens, int k0, double& D, double& Dz1, double& Dz2) {
    D = 0.0, Dz1 = 0.0, Dz2 = 0.0;
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
}
// The same three sums of a row whose 16 voxels have ONE type (nearly every row): d * (number of voxels), d * sum(2K + 1), d * sum(3K^2 + 3K
// + 1) — the integer sums from the row's bit mask (per byte of the mask a table holds the sum of the set bits' positions and of their
// squares), assembled in doubles that stay integers below 2^53, one product each. While the loop's sums are exact (grids of up to ~4000
// voxels along k for a density with a full mantissa; any size for a density like 1.0) this is the loop's result bit for bit, beyond that it
// rounds once where the loop rounds per term. Sixteen table reads, conversions and pairs of double multiply-adds less per row: the moment
// pass was a quarter of k_derive's vector instructions, and k_derive is bound by their issue.
// `tab`: moments_table_entry of every byte value.
__device__ __forceinline__ uint16_t moments_table_entry(uint32_t b) {
    uint32_t a1 = 0, a2 = 0;
    for (uint32_t k = 0; k < 8; ++k)
        if ((b >> k) & 1u) a1 += k, a2 += k * k;
    return (uint16_t)(a1 | (a2 << 8));  // <= 28, <= 140
}
__device__ __forceinline__ void moments_row_sums_one_type(uint32_t m, uint32_t type, const float* s_dens, const uint16_t* tab, int k0, double& D, double& Dz1,
                                                           double& Dz2) {
    const uint32_t lo = m & 0xFFu, hi = (m >> 8) & 0xFFu;
    const uint32_t tl = tab[lo], th = tab[hi], nh = __popc(hi);
    const uint32_t a1 = (tl & 0xFFu) + (th & 0xFFu) + 8u * nh;                       // sum of k
    const uint32_t a2 = (tl >> 8) + (th >> 8) + 16u * (th & 0xFFu) + 64u * nh;       // sum of k^2: (8 + k')^2 = 64 + 16 k' + k'^2 in the high byte
    const double d = (double)s_dens[type];
    const double S0 = (double)__popc(m), A1 = (double)a1, A2 = (double)a2, K0 = (double)k0;
    const double P1 = K0 * S0 + A1;                                                  // sum of K
    const double S1 = 2.0 * P1 + S0;
    const double S2 = (3.0 * (K0 * (K0 * S0 + 2.0 * A1) + A2) + 3.0 * P1) + S0;      // 3 sum K^2 + 3 sum K + n
    D = d * S0, Dz1 = d * S1, Dz2 = d * S2;
}

__device__ __forceinline__ void moments_reduce_rows(uint32_t tid, double D, double Dz1, double Dz2, double (*s_red)[10], int gi, int gj, double* __restrict__ out10);

__device__ __forceinline__ void chunk_moments_rows(uint32_t tid, uint32_t m, const uint32_t tw[4], const float* s_dens, double (*s_red)[10], int gi, int gj,
                                                   int k0, double* __restrict__ out10) {
    double D, Dz1, Dz2;
    moments_row_sums(m, tw, s_dens, k0, D, Dz1, Dz2);
    moments_reduce_rows(tid, D, Dz1, Dz2, s_red, gi, gj, out10);
}
// (with the byte table in LDS: the wave takes the one-type form unless one of its rows mixes types)
__device__ __forceinline__ void chunk_moments_rows_tab(uint32_t tid, uint32_t m, const uint32_t tw[4], const float* s_dens, const uint16_t* tab,
                                                       double (*s_red)[10], int gi, int gj, int k0, double* __restrict__ out10) {
    const uint32_t splat = (tw[0] & 0xFFu) * 0x01010101u;
    const bool mixed = ((tw[0] ^ splat) | (tw[1] ^ splat) | (tw[2] ^ splat) | (tw[3] ^ splat)) != 0u;
    double D, Dz1, Dz2;
    if (__builtin_amdgcn_ballot_w64(mixed) == 0ull) moments_row_sums_one_type(m, tw[0] & 0xFFu, s_dens, tab, k0, D, Dz1, Dz2);
    else moments_row_sums(m, tw, s_dens, k0, D, Dz1, Dz2);
    moments_reduce_rows(tid, D, Dz1, Dz2, s_red, gi, gj, out10);
}

__device__ __forceinline__ void moments_reduce_rows(uint32_t tid, double D, double Dz1, double Dz2, double (*s_red)[10], int gi, int gj, double* __restrict__ out10) {
    const double I = (double)gi, J = (double)gj;
    const double qx = 2.0 * I + 1.0, qy = 2.0 * J + 1.0;
    const double cx = 3.0 * I * I + 3.0 * I + 1.0, cy = 3.0 * J * J + 3.0 * J + 1.0;
    // The ten sums are linear in (D, Dz1, Dz2) with factors of i and of j. A DPP row of 16 lanes is one i: six row totals carry the
    // j-dependent parts (lane 15 of the row ends up with them), the i-dependent factors multiply the totals, and the 16 rows of the
    // workgroup meet in LDS — 72 DPP steps and 10 LDS words per wave where ten whole-wave sums took 120 steps and 80 lane reads
    // (this pass was half of k_derive's vector instructions, and those are what seven resident workgroups per CU queue for).
    const double r0 = ivx_row16_sum_f64(D), r1 = ivx_row16_sum_f64(D * qy), r2 = ivx_row16_sum_f64(D * cy);
    const double r3 = ivx_row16_sum_f64(Dz1), r4 = ivx_row16_sum_f64(qy * Dz1), r5 = ivx_row16_sum_f64(Dz2);
    if ((tid & 15u) == 15u) {
        double* o = s_red[tid >> 4];
        o[0] = r0;
        o[1] = r0 * qx;
        o[2] = r1;
        o[3] = r3;
        o[4] = r2 + r5;
        o[5] = r0 * cx + r5;
        o[6] = r0 * cx + r2;
        o[7] = r1 * qx;
        o[8] = r4;
        o[9] = r3 * qx;
    }
    __syncthreads();
    if (tid < 10) {
        double t = s_red[0][tid];
#pragma unroll
        for (int r = 1; r < 16; ++r) t += s_red[r][tid];  // fixed order: the same bits every time
        out10[tid] = t;
    }
}
