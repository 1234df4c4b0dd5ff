// Device-side bodies ("roles") of the Surface Nets kernels, shared by surface_nets.hip's stand-alone kernels and the fused step
// launches (step_fused.hip): a role is what one workgroup of the original kernel does, with its block index and block count passed
// in (`bid`, `nb`). Algorithm notes and reference citations: surface_nets.hip.
#pragma once
#include "ivx_internal.hpp"

namespace ivx_roles {
namespace sn {


constexpr int G = 18, NROWS = 324, NCROWS = 289, NCUBES = 4913;
// LDS tile: 18 x 18 rows of 18 bytes along k; a row occupies RS = 24 bytes with cell c at byte 3 + c, so the 16
// interior cells (c = 1..16, the chunk's own k-row) sit 4-byte aligned and are written as four words.
constexpr int RS = 24, TILE_BYTES = NROWS * RS;
constexpr uint32_t VPC = 864;  // vertices a mesher workgroup keeps in LDS for its quad phase (role_sn_emit)
__device__ __forceinline__ int tix(int a, int b, int c) { return (a * G + b) * RS + 3 + c; }

// (an empty asm the compiler must assume changes x: what is computed from the result cannot be hoisted out of the surrounding loop and held
// in registers across all of it — the mesher's per-thread row and cube-row constants were, and pushed the loads in flight out to scratch)
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}
struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 mul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float len3(V3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ __forceinline__ bool sneg(float f) { return (__float_as_uint(f) >> 31) != 0; }
// VoxelSignedDistance::to_f32 (lib.rs:220-222)
__device__ __forceinline__ float decode(int8_t e) { return (float)e * 0.02f; }

struct SnParams {
    GridView g;
    float extent;
    uint32_t x_off;
};

// What one padded row needs from memory, with every load ISSUED before any of them is used. Written the obvious way (record, then
// "if the neighbour is dense, its byte") the compiler sinks each load under the branch that consumes it and the row costs five
// dependent memory round trips; a tile is two rows per thread, i.e. ten round trips = most of the mesher's tile phase (measured:
// 7 us of a workgroup's 20). Here the addresses are clamped to valid ones, all loads go out back to back, `row_pin` keeps them from
// being sunk, and the records only select afterwards.
struct RowLoads {
    uint32_t c0, c1, c2;   // first word of the chunk records below / own / above along k (kind, generated kind, flags, uniform type)
    uint4 s4, t4;          // the 16 interior cells
    uint32_t b0s, b0t, b2s, b2t;  // k-halo bytes from the neighbours' face arrays
};
// 0: the row lies outside the grid (padding), 1: in the slab (RowLoads holds it), 2: ghost layer / other (serial path). Recomputed where it is
// needed instead of being carried beside the loads: a row in flight should hold registers for its data only.
__device__ __forceinline__ uint32_t row_mode(const GridView& g, int gi, int gj) {
    if (gj < 0 || gj >= (int)g.cy * 16) return 0u;
    if (gi < 0 || gi >= (int)g.cx * 16) return 2u;
    return 1u;
}
__device__ __forceinline__ void row_issue(const GridView& g, int gi, int gj, int ck, RowLoads& L) {
    if (row_mode(g, gi, gj) != 1u) return;
    const size_t cidx = (size_t)(((gi >> 4) * g.cy + (gj >> 4)) * g.cz) + (size_t)ck;
    const uint32_t rix = (uint32_t)(((gi & 15) << 4) | (gj & 15));
    const size_t o = (cidx << 12) + ((size_t)rix << 4);
    const bool has_lo = ck > 0, has_hi = ck + 1 < (int)g.cz;
    const uint32_t* ip = reinterpret_cast<const uint32_t*>(g.info + cidx);
    const uint8_t* kf = g.kface + cidx * 1024 + rix;
    const uint32_t* ip0 = has_lo ? ip - 2 : ip;
    const uint32_t* ip2 = has_hi ? ip + 2 : ip;
    const uint8_t* kf0 = has_lo ? kf - 1024 : kf;
    const uint8_t* kf2 = has_hi ? kf + 1024 : kf;
    L.c1 = *ip;
    L.s4 = *reinterpret_cast<const uint4*>(g.sdf + o);
    L.t4 = *reinterpret_cast<const uint4*>(g.type + o);
    L.c0 = *ip0;
    L.c2 = *ip2;
    L.b0s = kf0[256];
    L.b0t = kf0[768];
    L.b2s = kf2[0];
    L.b2t = kf2[512];
}
// (an empty asm that "modifies" the loaded registers: the loads cannot move below it)
__device__ __forceinline__ void row_pin(RowLoads& L) {
    asm volatile("" : "+v"(L.c0), "+v"(L.c1), "+v"(L.c2));
    asm volatile("" : "+v"(L.s4.x), "+v"(L.s4.y), "+v"(L.s4.z), "+v"(L.s4.w), "+v"(L.t4.x), "+v"(L.t4.y), "+v"(L.t4.z), "+v"(L.t4.w));
    asm volatile("" : "+v"(L.b0s), "+v"(L.b0t), "+v"(L.b2s), "+v"(L.b2t));
}
__device__ __forceinline__ ivx_chunk_info record_of(uint32_t w) {  // (the fields of the record's first word; the rest zero)
    ivx_chunk_info c;
    c.kind = (uint8_t)(w & 0xFFu);
    c.gen_kind = (uint8_t)((w >> 8) & 0xFFu);
    c.flags = (uint8_t)((w >> 16) & 0xFFu);
    c.uniform_type = (uint8_t)(w >> 24);
    c.face_dist = 0;
    c.region_count = 0;
    c.boundary_region_count = 0;
    return c;
}
__device__ __forceinline__ void fetch_row_serial(const GridView& g, int gi, int gj, int ck, uint32_t sd[6], uint32_t ty[6]);
// sd/ty: [0] = cell 0 (byte), [1..4] = the 16 interior cells (words), [5] = cell 17 (byte). Outside the grid / void -> (127, 255)
// (object/sdf.rs:410-508: void neighbours pad with +2.54). Inside the slab a Void / Uniform chunk is its record, not its planes.
__device__ __forceinline__ void row_finish(const GridView& g, int gi, int gj, int ck, const RowLoads& L, uint32_t sd[6], uint32_t ty[6]) {
    sd[0] = sd[5] = 0x7Fu;
    ty[0] = ty[5] = 0xFFu;
    sd[1] = sd[2] = sd[3] = sd[4] = 0x7F7F7F7Fu;
    ty[1] = ty[2] = ty[3] = ty[4] = 0xFFFFFFFFu;
    const uint32_t mode = row_mode(g, gi, gj);
    if (mode == 0u) return;
    if (mode == 2u) {
        fetch_row_serial(g, gi, gj, ck, sd, ty);
        return;
    }
    const ivx_chunk_info c1 = record_of(L.c1), c0 = record_of(L.c0), c2 = record_of(L.c2);
    const bool d1 = c1.kind == KIND_NONUNIFORM;
    const uint32_t us = ivx_uniform_sdf(c1.kind) * 0x01010101u, ut = ivx_uniform_type(c1) * 0x01010101u;
    sd[1] = d1 ? L.s4.x : us, sd[2] = d1 ? L.s4.y : us, sd[3] = d1 ? L.s4.z : us, sd[4] = d1 ? L.s4.w : us;
    ty[1] = d1 ? L.t4.x : ut, ty[2] = d1 ? L.t4.y : ut, ty[3] = d1 ? L.t4.z : ut, ty[4] = d1 ? L.t4.w : ut;
    if (ck > 0) {
        const bool dense = c0.kind == KIND_NONUNIFORM;
        sd[0] = dense ? L.b0s : ivx_uniform_sdf(c0.kind);
        ty[0] = dense ? L.b0t : ivx_uniform_type(c0);
    }
    if (ck + 1 < (int)g.cz) {
        const bool dense = c2.kind == KIND_NONUNIFORM;
        sd[5] = dense ? L.b2s : ivx_uniform_sdf(c2.kind);
        ty[5] = dense ? L.b2t : ivx_uniform_type(c2);
    }
}
__device__ __forceinline__ void fetch_row(const GridView& g, int gi, int gj, int ck, uint32_t sd[6], uint32_t ty[6]) {
    RowLoads L;
    row_issue(g, gi, gj, ck, L);
    if (row_mode(g, gi, gj) == 1u) row_pin(L);
    row_finish(g, gi, gj, ck, L, sd, ty);
}

// The same row by the plain dependent path: rows of a ghost layer (a slab's x halo) and, inside the slab, the reference form of the above.
__device__ __forceinline__ void fetch_row_serial(const GridView& g, int gi, int gj, int ck, uint32_t sd[6], uint32_t ty[6]) {
    // sd/ty: [0] = cell 0 (byte), [1..4] = the 16 interior cells (words), [5] = cell 17 (byte)
    sd[0] = sd[5] = 0x7Fu;
    ty[0] = ty[5] = 0xFFu;
    sd[1] = sd[2] = sd[3] = sd[4] = 0x7F7F7F7Fu;
    ty[1] = ty[2] = ty[3] = ty[4] = 0xFFFFFFFFu;
    if (gj < 0 || gj >= (int)g.cy * 16) return;
    const int8_t* ps;
    const uint8_t* pt;
    size_t row;        // offset of (.., gj, k = 0) inside the chunk / ghost column
    size_t kstride;    // offset between consecutive chunks along k
    if (gi < 0 || gi >= (int)g.cx * 16) {
        const int side = gi < 0 ? 0 : 1;
        if (!g.ghost_sdf[side]) return;
        ps = g.ghost_sdf[side];
        pt = g.ghost_type[side];
        row = (size_t)((gj >> 4) * g.cz) * 256 + ((gj & 15) << 4);
        kstride = 256;
    } else {
        ps = g.sdf;
        pt = g.type;
        row = ((size_t)(((gi >> 4) * g.cy + (gj >> 4)) * g.cz) << 12) + (((gi & 15) << 8) | ((gj & 15) << 4));
        kstride = IVX_CHUNK_VOXELS;
    }
    const size_t o = row + (size_t)ck * kstride;
    if (kstride == IVX_CHUNK_VOXELS) {
        // inside the slab a Void / Uniform chunk is its record, not its planes (compact planes)
        // (record and plane rows are fetched side by side and the record picks afterwards: the planes exist for every
        // chunk, only their content may be stale, and a dependent load chain costs more here than the extra bytes)
        const ivx_chunk_info* ip = g.info + (o >> 12);
        const ivx_chunk_info c1 = ip[0];
        const uint4 s4 = *reinterpret_cast<const uint4*>(ps + o);
        const uint4 t4 = *reinterpret_cast<const uint4*>(pt + o);
        const bool d1 = c1.kind == KIND_NONUNIFORM;
        const uint32_t us = ivx_uniform_sdf(c1.kind) * 0x01010101u, ut = ivx_uniform_type(c1) * 0x01010101u;
        sd[1] = d1 ? s4.x : us, sd[2] = d1 ? s4.y : us, sd[3] = d1 ? s4.z : us, sd[4] = d1 ? s4.w : us;
        ty[1] = d1 ? t4.x : ut, ty[2] = d1 ? t4.y : ut, ty[3] = d1 ? t4.z : ut, ty[4] = d1 ? t4.w : ut;
        // the k halo comes from the face bytes k_derive set side by side (kface): in the planes they sit 16 bytes apart, a cache
        // line for every four rows and plane — three times the lines of the tile's interior
        const uint8_t* kf = g.kface + (o >> 12) * 1024 + ((uint32_t)(o & 4095u) >> 4);
        if (ck > 0) {
            const ivx_chunk_info c0 = ip[-1];
            const uint32_t bs = kf[-1024 + 256], bt = kf[-1024 + 768];
            const bool dense = c0.kind == KIND_NONUNIFORM;
            sd[0] = dense ? bs : ivx_uniform_sdf(c0.kind);
            ty[0] = dense ? bt : ivx_uniform_type(c0);
        }
        if (ck + 1 < (int)g.cz) {
            const ivx_chunk_info c2 = ip[1];
            const uint32_t bs = kf[1024], bt = kf[1024 + 512];
            const bool dense = c2.kind == KIND_NONUNIFORM;
            sd[5] = dense ? bs : ivx_uniform_sdf(c2.kind);
            ty[5] = dense ? bt : ivx_uniform_type(c2);
        }
        return;
    }
    const uint4 s4 = *reinterpret_cast<const uint4*>(ps + o);
    const uint4 t4 = *reinterpret_cast<const uint4*>(pt + o);
    sd[1] = s4.x, sd[2] = s4.y, sd[3] = s4.z, sd[4] = s4.w;
    ty[1] = t4.x, ty[2] = t4.y, ty[3] = t4.z, ty[4] = t4.w;
    if (ck > 0) {
        sd[0] = (uint8_t)ps[o - kstride + 15];
        ty[0] = pt[o - kstride + 15];
    }
    if (ck + 1 < (int)g.cz) {
        sd[5] = (uint8_t)ps[o + kstride];
        ty[5] = pt[o + kstride];
    }
}

// sign bits of 16 packed i8 -> 16-bit mask
__device__ __forceinline__ uint32_t neg16(const uint32_t w[4]) {
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t sb = w[q] & 0x80808080u;
        m |= (((sb >> 7) & 1u) | ((sb >> 14) & 2u) | ((sb >> 21) & 4u) | ((sb >> 28) & 8u)) << (4 * q);
    }
    return m;
}

// The 18 sign bits of one padded row from the per-row masks k_derive left (count pass): bit 0 = cell -1 (bit 15 of the row in
// the chunk below along k), bits 1..16 the row itself, bit 17 = cell 16 (bit 0 of the chunk above). A chunk that is not
// NonUniform has no masks: all negative when Uniform, none when Void. Rows of a ghost layer come from the ghost planes.
// (split like RowLoads: every load issued, then pinned, then the records select)
struct SignLoads {
    uint32_t k0, k1, k2, m0, m1, m2;
    uint32_t mode;  // 0: outside the grid, 1: in the slab, 2: ghost layer (serial path)
    uint32_t has_lo, has_hi;
};
__device__ __forceinline__ void signs_issue(const GridView& g, int gi, int gj, int ck, SignLoads& L) {
    L.mode = 0u;
    if (gj < 0 || gj >= (int)g.cy * 16) return;
    if (gi < 0 || gi >= (int)g.cx * 16) {
        L.mode = 2u;
        return;
    }
    L.mode = 1u;
    const uint32_t chunk = ((gi >> 4) * g.cy + (gj >> 4)) * g.cz + ck;
    const uint32_t row = ((gi & 15) << 4) | (gj & 15);
    L.has_lo = ck > 0 ? 1u : 0u;
    L.has_hi = ck + 1 < (int)g.cz ? 1u : 0u;
    const ivx_chunk_info* ip = g.info + chunk;
    const uint16_t* sp = g.signs + (size_t)chunk * 256 + row;
    const ivx_chunk_info* ip0 = L.has_lo ? ip - 1 : ip;
    const ivx_chunk_info* ip2 = L.has_hi ? ip + 1 : ip;
    const uint16_t* sp0 = L.has_lo ? sp - 256 : sp;
    const uint16_t* sp2 = L.has_hi ? sp + 256 : sp;
    L.k1 = ip->kind;
    L.m1 = *sp;
    L.k0 = ip0->kind;
    L.m0 = *sp0;
    L.k2 = ip2->kind;
    L.m2 = *sp2;
}
__device__ __forceinline__ void signs_pin(SignLoads& L) { asm volatile("" : "+v"(L.k0), "+v"(L.k1), "+v"(L.k2), "+v"(L.m0), "+v"(L.m1), "+v"(L.m2)); }
__device__ __forceinline__ uint32_t signs_finish(const GridView& g, int gi, int gj, int ck, const SignLoads& L) {
    if (L.mode == 0u) return 0u;
    if (L.mode == 2u) {
        uint32_t sd[6], ty[6];
        fetch_row(g, gi, gj, ck, sd, ty);
        return ((sd[0] >> 7) & 1u) | (neg16(sd + 1) << 1) | (((sd[5] >> 7) & 1u) << 17);
    }
    uint32_t bits = (L.k1 == KIND_NONUNIFORM ? L.m1 : (L.k1 == KIND_UNIFORM ? 0xFFFFu : 0u)) << 1;
    if (L.has_lo) bits |= L.k0 == KIND_NONUNIFORM ? ((L.m0 >> 15) & 1u) : (L.k0 == KIND_UNIFORM ? 1u : 0u);
    if (L.has_hi) bits |= (L.k2 == KIND_NONUNIFORM ? (L.m2 & 1u) : (L.k2 == KIND_UNIFORM ? 1u : 0u)) << 17;
    return bits;
}

__device__ __forceinline__ uint32_t neighbour_kind(const GridView& g, int ci, int cj, int ck) {
    if (cj < 0 || ck < 0 || cj >= (int)g.cy || ck >= (int)g.cz) return KIND_VOID;
    if (ci < 0) return g.ghost_info[0] ? g.ghost_info[0][cj * g.cz + ck].kind : (uint32_t)KIND_VOID;
    if (ci >= (int)g.cx) return g.ghost_info[1] ? g.ghost_info[1][cj * g.cz + ck].kind : (uint32_t)KIND_VOID;
    return g.info[(ci * g.cy + cj) * g.cz + ck].kind;
}

// Stage the 18^3 padded tile: 324 rows, one 16-byte plane load each (+ two halo bytes). s_neg[r] = 18-bit mask of
// negative distances (decoded 0 is +0.0 => outside, surface_nets.rs:209-224). s_sd / s_ty may be null (count pass).
__device__ __forceinline__ void load_tile(const GridView& g, int ci, int cj, int ck, uint8_t* s_sd, uint8_t* s_ty, uint32_t* s_neg, uint32_t tid) {
    constexpr int ROUNDS = (NROWS + 255) / 256;
    if (!s_sd) {  // count pass: signs only
        SignLoads S[ROUNDS];
#pragma unroll
        for (int it = 0; it < ROUNDS; ++it) {
            const int r = (int)tid + 256 * it;
            S[it].mode = 0u;
            if (r < NROWS) {
                const int a = r / G, b = r - a * G;
                signs_issue(g, ci * 16 + a - 1, cj * 16 + b - 1, ck, S[it]);
            }
        }
#pragma unroll
        for (int it = 0; it < ROUNDS; ++it)
            if (S[it].mode == 1u) signs_pin(S[it]);
#pragma unroll
        for (int it = 0; it < ROUNDS; ++it) {
            const int r = (int)tid + 256 * it;
            if (r >= NROWS) break;
            const int a = r / G, b = r - a * G;
            s_neg[r] = signs_finish(g, ci * 16 + a - 1, cj * 16 + b - 1, ck, S[it]);
        }
        return;
    }
    // the loads of all of the thread's rows go out before the first of them is looked at (see RowLoads)
    RowLoads L[ROUNDS];
#pragma unroll
    for (int it = 0; it < ROUNDS; ++it) {
        const int r = (int)tid + 256 * it;
        if (r < NROWS) {
            const int a = r / G, b = r - a * G;
            row_issue(g, ci * 16 + a - 1, cj * 16 + b - 1, ck, L[it]);
        }
    }
#pragma unroll
    for (int it = 0; it < ROUNDS; ++it) {
        const int r = (int)tid + 256 * it;
        const int a = r / G, b = r - a * G;
        if (r < NROWS && row_mode(g, ci * 16 + a - 1, cj * 16 + b - 1) == 1u) row_pin(L[it]);
    }
#pragma unroll
    for (int it = 0; it < ROUNDS; ++it) {
        const int r = (int)tid + 256 * it;
        if (r >= NROWS) break;
        const int a = r / G, b = r - a * G;
        uint32_t sd[6], ty[6];
        row_finish(g, ci * 16 + a - 1, cj * 16 + b - 1, ck, L[it], sd, ty);
        s_neg[r] = ((sd[0] >> 7) & 1u) | (neg16(sd + 1) << 1) | (((sd[5] >> 7) & 1u) << 17);
        uint8_t* ds = s_sd + r * RS;
        uint8_t* dt = s_ty + r * RS;
        ds[3] = (uint8_t)sd[0];
        dt[3] = (uint8_t)ty[0];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            reinterpret_cast<uint32_t*>(ds + 4)[q] = sd[1 + q];
            reinterpret_cast<uint32_t*>(dt + 4)[q] = ty[1 + q];
        }
        ds[20] = (uint8_t)sd[5];
        dt[20] = (uint8_t)ty[5];
    }
}

// Ordered block-wide exclusive prefix of `val` in thread order; `total` = block sum (same in every thread).
// (TRAILING = false: no barrier behind the read of the wave sums — for a caller that does not write them again before another barrier)
template <bool TRAILING = true>
__device__ __forceinline__ uint32_t block_prefix(uint32_t val, uint32_t* s_wsum, uint32_t tid, uint32_t& total) {
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
    if (TRAILING) __syncthreads();
    return wbase + incl - val;
}

// Bit k of the results refers to the cube (i, j, k) of cube row cr = i*17 + j (k = 0..16):
//   vbits: the cube holds a vertex (its 8 corner signs are mixed, surface_nets.rs:209-224)
//   qx/qy/qz: it emits the quad of its X / Y / Z edge (maybe_make_surface_nets_quad, surface_nets.rs:251-334)
__device__ __forceinline__ void cube_row_bits(const uint32_t* s_neg, int i, int j, const int* upper, uint32_t& vbits, uint32_t& qx, uint32_t& qy,
                                              uint32_t& qz) {
    const uint32_t r00 = s_neg[i * G + j], r01 = s_neg[i * G + j + 1], r10 = s_neg[(i + 1) * G + j], r11 = s_neg[(i + 1) * G + j + 1];
    const uint32_t o = r00 | r01 | r10 | r11, a = r00 & r01 & r10 & r11;
    vbits = ((o | (o >> 1)) & ~(a & (a >> 1))) & 0x1FFFFu;
    const uint32_t knz = 0x1FFFEu;  // k != 0
    qx = (j != 0 && i < upper[0]) ? ((r00 ^ r10) & knz) : 0u;
    qy = (i != 0 && j < upper[1]) ? ((r00 ^ r01) & knz) : 0u;
    qz = (i != 0 && j != 0) ? ((r00 ^ (r00 >> 1)) & ((1u << upper[2]) - 1u)) : 0u;
}

__device__ __forceinline__ bool chunk_exposed(const ivx_chunk_info& ci) {
    return ci.kind == KIND_NONUNIFORM && (ci.flags & CF_FULLY_OBSCURED) != CF_FULLY_OBSCURED;
}

// (issued before the tile's loads and looked at after them: the three records ride along instead of costing round trips of their own)
struct UpperLoads {
    uint32_t k[3];
};
__device__ __forceinline__ void upper_issue(const GridView& g, int ci, int cj, int ck, UpperLoads& U) {
    const uint32_t own = (uint32_t)((ci * (int)g.cy + cj) * (int)g.cz + ck);
    const ivx_chunk_info* px = g.info + own;  // clamped to the chunk itself where there is no neighbour record to read
    bool gx = false;
    if (ci + 1 < (int)g.cx) px = g.info + own + g.cy * g.cz;
    else if (g.ghost_info[1]) px = g.ghost_info[1] + (cj * (int)g.cz + ck), gx = true;
    const ivx_chunk_info* py = cj + 1 < (int)g.cy ? g.info + own + g.cz : g.info + own;
    const ivx_chunk_info* pz = ck + 1 < (int)g.cz ? g.info + own + 1 : g.info + own;
    // the records' first words, untouched (kind = low byte, taken in upper_finish): an operation on a loaded value here would wait for the load
    // here; and through an index that is zero but not to the compiler, so that they are not moved to scalar registers at once either
    const uint32_t z = opaque(0u);
    U.k[0] = reinterpret_cast<const uint32_t*>(px)[z];
    U.k[1] = reinterpret_cast<const uint32_t*>(py)[z];
    U.k[2] = reinterpret_cast<const uint32_t*>(pz)[z];
    (void)gx;
}
template <bool PIN = true>
__device__ __forceinline__ void upper_finish(const GridView& g, int ci, int cj, int ck, UpperLoads& U, int* upper) {
    // the upper layer of cubes belongs to the upper neighbour chunk when that chunk is non-uniform (surface_nets.rs:252-261)
    if (PIN) asm volatile("" : "+v"(U.k[0]), "+v"(U.k[1]), "+v"(U.k[2]));
    const bool hx = ci + 1 < (int)g.cx || g.ghost_info[1] != nullptr, hy = cj + 1 < (int)g.cy, hz = ck + 1 < (int)g.cz;
    upper[0] = upper[1] = upper[2] = G - 1;
    if (hx && (U.k[0] & 0xFFu) == KIND_NONUNIFORM) upper[0] -= 1;
    if (hy && (U.k[1] & 0xFFu) == KIND_NONUNIFORM) upper[1] -= 1;
    if (hz && (U.k[2] & 0xFFu) == KIND_NONUNIFORM) upper[2] -= 1;
}

// Walks the active list (the chunks k_chunk_pre settled have no mesh and got their zero counts there).
__device__ __forceinline__ void role_sn_count(uint32_t bid, uint32_t nb, SnParams p, uint32_t* __restrict__ counts, uint32_t* __restrict__ group_sums,

                                                  const uint32_t* __restrict__ work_counts, const uint32_t* __restrict__ active_list) {
    __shared__ uint32_t s_neg[NROWS];
    __shared__ uint32_t s_acc[2];
    const GridView& g = p.g;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_active = work_counts[0];
    for (uint32_t li = ivx_xcd_remap(bid, nb); li < n_active; li += nb) {
    __syncthreads();  // the previous chunk's LDS use is over
    const uint32_t entry = active_list[li];
    const uint32_t chunk = IVX_LIST_CHUNK(entry);
    if (!IVX_LIST_EXPOSED(entry)) {
        if (tid == 0) {
            counts[2 * chunk] = 0;
            counts[2 * chunk + 1] = 0;
        }
        continue;
    }
    const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
    if (tid < 2) s_acc[tid] = 0;
    UpperLoads ul;
    upper_issue(g, ci, cj, ck, ul);
    load_tile(g, ci, cj, ck, nullptr, nullptr, s_neg, tid);
    int upper[3];
    upper_finish(g, ci, cj, ck, ul, upper);
    __syncthreads();
    uint32_t nv = 0, nq = 0;
    for (int cr = tid; cr < NCROWS; cr += 256) {
        uint32_t vb, qx, qy, qz;
        cube_row_bits(s_neg, cr / 17, cr % 17, upper, vb, qx, qy, qz);
        nv += __popc(vb);
        nq += __popc(qx) + __popc(qy) + __popc(qz);
    }
    {  // wave totals in registers, one LDS add per wave (same-address LDS atomics of many lanes are very slow)
        const uint32_t wv = ivx_wave_sum(nv), wq = ivx_wave_sum(nq);
        if ((tid & 63u) == 0) {
            atomicAdd(&s_acc[0], wv);
            atomicAdd(&s_acc[1], wq);
        }
    }
    __syncthreads();
    if (tid == 0) {
        counts[2 * chunk] = s_acc[0];
        counts[2 * chunk + 1] = s_acc[1] * 6u;
        if (s_acc[1]) {  // first level of the scan over chunks: totals per group of 256 chunks (vertices, indices, submeshes)
            uint32_t* gs = group_sums + 3 * (chunk >> 8);
            atomicAdd(gs, s_acc[0]);
            atomicAdd(gs + 1, s_acc[1] * 6u);
            atomicAdd(gs + 2, 1u);
        }
    }
    }
}

// After an edit: which chunks of the touched box grown by one chunk each way (the box the derive sweep has just gone over) have a mesh to
// renew — handle_chunk_voxels_modified (object/intersection.rs:560-598): a touched chunk, and a neighbour when the touched voxel range of
// the chunk comes within two voxels of the face they share —, and what their meshes need now (the count pass of the remesh for just these
// chunks). One workgroup per chunk of the grown box; out[4 b ..]: bit 31 = invalidated | flags << 8 | kind, vertices, indices, 0. With this
// in the edit's result block ivx_mesh_sync has its sizes without a count pass over the object and without a round trip of its own.
struct BoxNeeds {
    uint32_t t_lo[3], t_cc[3];  // the touched box (chunks) the ranges are indexed by
    uint32_t b_lo[3], b_cc[3];  // the grown box
};
// (`list` != null: the chunks named there instead of a box, every one of them taken as invalidated — ivx_mesh_sync's own count when the
// invalidated set is not the last edit's)
__device__ __forceinline__ void role_box_mesh_needs(uint32_t b, const SnParams& p, const BoxNeeds& bx, const uint32_t* __restrict__ touched,
                                                    const uint32_t* __restrict__ list, uint32_t* __restrict__ out, uint32_t* lds) {
    uint32_t* s_neg = lds;           // [NROWS]
    uint32_t* s_acc = lds + NROWS;   // [2]
    const GridView& g = p.g;
    const uint32_t tid = threadIdx.x;
    uint32_t bk = b % bx.b_cc[2], bj = (b / bx.b_cc[2]) % bx.b_cc[1], bi = b / (bx.b_cc[2] * bx.b_cc[1]);
    if (list) {
        const uint32_t c = list[b];
        bk = c % g.cz, bj = (c / g.cz) % g.cy, bi = c / (g.cz * g.cy);
    }
    const uint32_t idx[3] = {bx.b_lo[0] + bi, bx.b_lo[1] + bj, bx.b_lo[2] + bk};
    auto word = [&](int di, int dj, int dk) -> uint32_t {  // the touched word of the chunk at idx + d; 0 outside the touched box
        const uint32_t a[3] = {idx[0] + (uint32_t)di - bx.t_lo[0], idx[1] + (uint32_t)dj - bx.t_lo[1], idx[2] + (uint32_t)dk - bx.t_lo[2]};
        if (a[0] >= bx.t_cc[0] || a[1] >= bx.t_cc[1] || a[2] >= bx.t_cc[2]) return 0u;  // (unsigned: also below the box)
        return touched[(a[0] * bx.t_cc[1] + a[1]) * bx.t_cc[2] + a[2]];
    };
    bool inval = list != nullptr || word(0, 0, 0) != 0u;
#pragma unroll
    for (int d = 0; d < 3 && !list; ++d) {
        const uint32_t up = word(d == 0, d == 1, d == 2), dn = word(-(d == 0), -(d == 1), -(d == 2));
        if (up && ((up >> (4 * d)) & 15u) < 2u) inval = true;                     // the upper neighbour's touched range starts within two voxels of our face
        if (dn && 16u - (((dn >> (12 + 4 * d)) & 15u) + 1u) < 2u) inval = true;  // the lower neighbour's ends within two voxels of it
    }
    const uint32_t chunk = (idx[0] * g.cy + idx[1]) * g.cz + idx[2];
    if (!inval) {
        if (tid == 0) out[4 * b] = 0u, out[4 * b + 1] = 0u, out[4 * b + 2] = 0u, out[4 * b + 3] = chunk;
        return;
    }
    const ivx_chunk_info rec = g.info[chunk];
    const bool exposed = chunk_exposed(rec);
    uint32_t nv = 0, nq = 0;
    if (exposed) {  // (workgroup-uniform)
        const int ci = (int)idx[0], cj = (int)idx[1], ck = (int)idx[2];
        if (tid < 2) s_acc[tid] = 0;
        UpperLoads ul;
        upper_issue(g, ci, cj, ck, ul);
        load_tile(g, ci, cj, ck, nullptr, nullptr, s_neg, tid);
        int upper[3];
        upper_finish(g, ci, cj, ck, ul, upper);
        __syncthreads();
        for (int cr = tid; cr < NCROWS; cr += 256) {
            uint32_t vb, qx, qy, qz;
            cube_row_bits(s_neg, cr / 17, cr % 17, upper, vb, qx, qy, qz);
            nv += __popc(vb);
            nq += __popc(qx) + __popc(qy) + __popc(qz);
        }
        const uint32_t wv = ivx_wave_sum(nv), wq = ivx_wave_sum(nq);
        if ((tid & 63u) == 0) {
            atomicAdd(&s_acc[0], wv);
            atomicAdd(&s_acc[1], wq);
        }
        __syncthreads();
        nv = s_acc[0], nq = s_acc[1];
    }
    if (tid == 0) {
        out[4 * b] = 0x80000000u | ((uint32_t)rec.flags << 8) | (uint32_t)rec.kind;
        out[4 * b + 1] = nv;
        out[4 * b + 2] = nq * 6u;
        out[4 * b + 3] = chunk;
    }
}

// The count pass with one WAVE per listed chunk (four chunks per workgroup, no workgroup barrier): the pass is a chain of two trips to memory
// (list entry, then the sign rows) around a few hundred instructions, so what it needs is chunks in flight — 32 per CU this way against 8
// with a workgroup per chunk. Lanes 0..26 fetch the first word of the 27 chunk records of the tile's neighbourhood and the rows take their
// three kinds from those lanes (ds_bpermute) instead of loading them per row; a lane holds the three 16-bit sign masks of each of its six rows.
// `lds`: 4 x NROWS words of the caller's LDS (the fused launch lends the exact-numbering role's block: a role of its own LDS on top would push
// the launch past the eight workgroups per CU its register budget is set for).
__device__ __forceinline__ void role_sn_count_waves(uint32_t bid, uint32_t nb, SnParams p, uint32_t* __restrict__ counts, uint32_t* __restrict__ group_sums,
                                                    const uint32_t* __restrict__ work_counts, const uint32_t* __restrict__ active_list, uint32_t* lds, uint32_t COUNT_RUN,
                                                    uint32_t x_part = 0u) {
    constexpr int WC = (NCROWS + 63) / 64;
    const GridView& g = p.g;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* s_neg = lds + wave * NROWS;
    const uint32_t n_active = work_counts[0];
    // A wave takes runs of COUNT_RUN consecutive list entries and adds a run's totals to the group sums once per group it met, not once per
    // chunk: 256 consecutive chunks share the three words of a group, and three atomics per chunk on one cache line — 97 000 on the all-surface
    // 512^3 grid — were what the pass took its 0.22 ms for, whatever the rest of it did.
    uint32_t acc_group = 0xFFFFFFFFu, acc_v = 0u, acc_i = 0u, acc_s = 0u;  // (lane 0)
    auto flush = [&]() {
        if (lane == 0 && acc_group != 0xFFFFFFFFu) {
            uint32_t* gs = group_sums + IVX_SN_GROUP_WORDS * acc_group;
            atomicAdd(gs, acc_v);
            atomicAdd(gs + 1, acc_i);
            atomicAdd(gs + 2, acc_s);
        }
        acc_v = acc_i = acc_s = 0u;
    };
    for (uint32_t run = ivx_xcd_remap(bid, nb) * 4u + wave; run * COUNT_RUN < n_active; run += nb * 4u)
    for (uint32_t li = run * COUNT_RUN; li < run * COUNT_RUN + COUNT_RUN && li < n_active; ++li) {
        const uint32_t entry = active_list[li];
        const uint32_t chunk = IVX_LIST_CHUNK(entry);
        if (ivx_xpart_skip(g, x_part, chunk / (g.cz * g.cy))) continue;  // (the other part's chunk)
        if (!IVX_LIST_EXPOSED(entry)) {
            if (lane == 0) reinterpret_cast<uint2*>(counts)[chunk] = make_uint2(0u, 0u);
            continue;
        }
        const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
        // the neighbourhood's records (a chunk that is not there reads as Void: kind 0)
        uint32_t rec = 0u;
        if (lane < 27u) {
            const int ni = ci + (int)(lane / 9u) - 1, nj = cj + (int)((lane / 3u) % 3u) - 1, nk = ck + (int)(lane % 3u) - 1;
            if (nj >= 0 && nj < (int)g.cy && nk >= 0 && nk < (int)g.cz) {
                const ivx_chunk_info* rp = nullptr;
                if (ni >= 0 && ni < (int)g.cx) rp = g.info + (size_t)((ni * (int)g.cy + nj) * (int)g.cz + nk);
                else if (g.ghost_info[ni < 0 ? 0 : 1]) rp = g.ghost_info[ni < 0 ? 0 : 1] + (nj * (int)g.cz + nk);
                if (rp) rec = *reinterpret_cast<const uint32_t*>(rp);
            }
        }
        // Every load is issued before any is looked at. The chunk's OWN 256 rows first, four per lane: row = lane + 64 t is row (i, j) =
        // (row >> 4, row & 15) of the chunk and of the chunks below / above along k, at a uniform base + 2 row — no per-row chunk arithmetic,
        // and their three kinds are the same for all (scalars). Then the 68 rows of the tile's rim, which belong to eight other chunk
        // columns: the general form, two turns (64 + 4 rows).
        const uint16_t* own = g.signs + (size_t)chunk * 256;
        const uint16_t* own_lo = ck > 0 ? own - 256 : own;  // (clamped to the chunk itself where there is no neighbour; not consulted then)
        const uint16_t* own_hi = ck + 1 < (int)g.cz ? own + 256 : own;
        uint32_t a0[4], a1[4], a2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t row = lane + 64u * (uint32_t)t;
            a1[t] = own[row];
            a0[t] = own_lo[row];
            a2[t] = own_hi[row];
        }
        uint32_t m0[2], m1[2], m2[2];
        int ra[2], rb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int h = (int)lane + 64 * t;  // rim row h: the two i-faces (18 rows each), then the two j-faces (16 + 16, interleaved)
            ra[t] = h < 18 ? 0 : (h < 36 ? G - 1 : 1 + ((h - 36) >> 1));
            rb[t] = h < 18 ? h : (h < 36 ? h - 18 : ((h & 1) ? G - 1 : 0));
            int gi = ci * 16 + ra[t] - 1, gj = cj * 16 + rb[t] - 1;
            const bool real = h < 68 && row_mode(g, gi, gj) == 1u;
            gi = real ? gi : ci * 16;
            gj = real ? gj : cj * 16;
            const uint32_t ch = ((gi >> 4) * g.cy + (gj >> 4)) * g.cz + ck;
            const uint16_t* sp = g.signs + (size_t)ch * 256 + (((gi & 15) << 4) | (gj & 15));
            m1[t] = *sp;
            m0[t] = *(ck > 0 ? sp - 256 : sp);
            m2[t] = *(ck + 1 < (int)g.cz ? sp + 256 : sp);
        }
        {
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)rec, 12) & 0xFFu, k1 = (uint32_t)__builtin_amdgcn_readlane((int)rec, 13) & 0xFFu,
                           k2 = (uint32_t)__builtin_amdgcn_readlane((int)rec, 14) & 0xFFu;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t row = lane + 64u * (uint32_t)t;
                uint32_t bits = (k1 == KIND_NONUNIFORM ? a1[t] : (k1 == KIND_UNIFORM ? 0xFFFFu : 0u)) << 1;
                if (ck > 0) bits |= k0 == KIND_NONUNIFORM ? ((a0[t] >> 15) & 1u) : (k0 == KIND_UNIFORM ? 1u : 0u);
                if (ck + 1 < (int)g.cz) bits |= (k2 == KIND_NONUNIFORM ? (a2[t] & 1u) : (k2 == KIND_UNIFORM ? 1u : 0u)) << 17;
                s_neg[((row >> 4) + 1u) * G + (row & 15u) + 1u] = bits;
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int h = (int)lane + 64 * t;
            const int gi = ci * 16 + ra[t] - 1, gj = cj * 16 + rb[t] - 1;
            const uint32_t mode = h < 68 ? row_mode(g, gi, gj) : 0u;
            // (every lane takes part in the exchange; the column of a row that is not fetched is the chunk's own)
            const int col = mode == 1u ? (((gi >> 4) - ci + 1) * 3 + ((gj >> 4) - cj + 1)) * 3 : 12;
            const uint32_t k0 = (uint32_t)__shfl((int)rec, col, 64) & 0xFFu, k1 = (uint32_t)__shfl((int)rec, col + 1, 64) & 0xFFu,
                           k2 = (uint32_t)__shfl((int)rec, col + 2, 64) & 0xFFu;
            uint32_t bits = 0u;
            if (mode == 1u) {
                bits = (k1 == KIND_NONUNIFORM ? m1[t] : (k1 == KIND_UNIFORM ? 0xFFFFu : 0u)) << 1;
                if (ck > 0) bits |= k0 == KIND_NONUNIFORM ? ((m0[t] >> 15) & 1u) : (k0 == KIND_UNIFORM ? 1u : 0u);
                if (ck + 1 < (int)g.cz) bits |= (k2 == KIND_NONUNIFORM ? (m2[t] & 1u) : (k2 == KIND_UNIFORM ? 1u : 0u)) << 17;
            } else if (mode == 2u) {  // a row of a ghost layer: from the ghost planes
                uint32_t sd[6], ty[6];
                fetch_row_serial(g, gi, gj, ck, sd, ty);
                bits = ((sd[0] >> 7) & 1u) | (neg16(sd + 1) << 1) | (((sd[5] >> 7) & 1u) << 17);
            }
            if (h < 68) s_neg[ra[t] * G + rb[t]] = bits;
        }
        // the upper layer of cubes belongs to the upper neighbour chunk when that chunk is non-uniform (surface_nets.rs:252-261)
        int upper[3] = {G - 1, G - 1, G - 1};
        if (((uint32_t)__shfl((int)rec, 22, 64) & 0xFFu) == KIND_NONUNIFORM) upper[0] -= 1;
        if (((uint32_t)__shfl((int)rec, 16, 64) & 0xFFu) == KIND_NONUNIFORM) upper[1] -= 1;
        if (((uint32_t)__shfl((int)rec, 14, 64) & 0xFFu) == KIND_NONUNIFORM) upper[2] -= 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t nv = 0, nq = 0;
#pragma unroll
        for (int it = 0; it < WC; ++it) {
            const int cr = (int)lane + 64 * it;
            if (cr < NCROWS) {
                uint32_t vb, qx, qy, qz;
                cube_row_bits(s_neg, cr / 17, cr % 17, upper, vb, qx, qy, qz);
                nv += __popc(vb);
                nq += __popc(qx) + __popc(qy) + __popc(qz);
            }
        }
        const uint32_t wv = ivx_wave_sum(nv), wq = ivx_wave_sum(nq);
        if (lane == 0) {
            reinterpret_cast<uint2*>(counts)[chunk] = make_uint2(wv, wq * 6u);
        }
        if (wq) {  // first level of the scan over chunks: totals per group of 256 chunks (vertices, indices, submeshes)
            if ((chunk >> 8) != acc_group) {
                flush();
                acc_group = chunk >> 8;
            }
            acc_v += wv, acc_i += wq * 6u, acc_s += 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the rows in LDS are rewritten for the wave's next chunk)
    }
    flush();
}

// consecutive list entries a wave of the count pass takes per turn: 1 while that still fills the chip (32 waves per CU), more for longer lists
static inline uint32_t ivx_count_run(const ivx_grid* g) {
    const uint32_t r = ivx_list_grid(g) / ((uint32_t)g->ctx->n_cu * 32u);
    return r < 1u ? 1u : (r > 8u ? 8u : r);
}

// Exclusive scan over chunks in chunk-linear order of (vertices, indices, submesh) with chunks whose
// index count is zero contributing nothing (mesh.rs:321-323). offsets[2c], offsets[2c+1]; totals at
// offsets[2n..2n+3); submesh rank at ranks[c].
// `walk` (the fused step path; null elsewhere): the order the mesher's main pass takes the list in — the chunks with many vertices first. Its
// workgroups draw entries as they go, a thousand at a time over a list three to four times that, and a chunk takes 4 to 20 us: with the list in
// chunk order the launch's last quarter is a few workgroups finishing whatever long entries came up last. Two classes are enough to end
// on short ones: entries of at least `walk_big` vertices are listed from the front of `walk`, the others from its back, records (walk[0 .. n))
// and list indices (walk_li) side by side so that a draw is still two independent loads; the class counts in walk_count[0 / 1] (zero at the start of a step).
struct SnWalk {
    uint4* items;         // [n_chunks]
    uint32_t* li;         // [n_chunks]
    uint32_t* count;      // [2]: entries listed from the front, from the back
    uint32_t big;         // vertices from which an entry is listed from the front
};
__device__ __forceinline__ void role_sn_scan(uint32_t bid, uint32_t nb, uint32_t n_chunks, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ group_sums,
                                                 uint32_t* __restrict__ offsets, uint32_t* __restrict__ ranks, uint4* __restrict__ emit_items, SnWalk walk = SnWalk{nullptr, nullptr, nullptr, 0u}) {
    // block b = chunks [256 b, 256 b + 256): base = totals of the groups before it, then an ordered block prefix
    __shared__ uint32_t s_w[3][4];
    __shared__ uint32_t s_base[3];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t p0 = 0, p1 = 0, p2 = 0;
    for (uint32_t b = tid; b < bid; b += 256u) {
        p0 += group_sums[IVX_SN_GROUP_WORDS * b];
        p1 += group_sums[IVX_SN_GROUP_WORDS * b + 1];
        p2 += group_sums[IVX_SN_GROUP_WORDS * b + 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        p0 += __shfl_down(p0, o, 64);
        p1 += __shfl_down(p1, o, 64);
        p2 += __shfl_down(p2, o, 64);
    }
    if (lane == 0) {
        s_w[0][wave] = p0;
        s_w[1][wave] = p1;
        s_w[2][wave] = p2;
    }
    __syncthreads();
    if (tid < 3) s_base[tid] = (s_w[tid][0] + s_w[tid][1]) + (s_w[tid][2] + s_w[tid][3]);
    __syncthreads();
    const uint32_t b0 = s_base[0], b1 = s_base[1], b2 = s_base[2];
    const uint32_t c = bid * 256u + tid;
    uint2 vi = make_uint2(0u, 0u);
    if (c < n_chunks) vi = reinterpret_cast<const uint2*>(counts)[c];
    const uint32_t on = vi.y != 0u;
    const uint32_t sv = on ? vi.x : 0u, si = vi.y, ss = on;
    uint32_t iv = sv, ii = si, is = ss;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a0 = __shfl_up(iv, o, 64), a1 = __shfl_up(ii, o, 64), a2 = __shfl_up(is, o, 64);
        if (lane >= (uint32_t)o) {
            iv += a0;
            ii += a1;
            is += a2;
        }
    }
    __syncthreads();
    if (lane == 63u) {
        s_w[0][wave] = iv;
        s_w[1][wave] = ii;
        s_w[2][wave] = is;
    }
    __syncthreads();
    uint32_t wv = 0, wi = 0, ws = 0, tv = 0, ti = 0, ts = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
        if (w < wave) {
            wv += s_w[0][w];
            wi += s_w[1][w];
            ws += s_w[2][w];
        }
        tv += s_w[0][w];
        ti += s_w[1][w];
        ts += s_w[2][w];
    }
    if (c < n_chunks) {
        reinterpret_cast<uint2*>(offsets)[c] = make_uint2(b0 + wv + iv - sv, b1 + wi + ii - si);
        ranks[c] = b2 + ws + is - ss;
        // the chunks with a mesh, in chunk order, for k_sn_emit (a list built here costs nothing; appending to it from the
        // count pass meant thousands of returning atomics on one address)
        // one record per meshed chunk: chunk, vertex offset, index offset, vertex count | quads << 16 — everything the emit
        // pass needs to start loading its tile after a single fetch
        if (on) emit_items[b2 + ws + is - ss] = make_uint4(c, b0 + wv + iv - sv, b1 + wi + ii - si, vi.x | ((vi.y / 6u) << 16));
    }
    if (walk.items) {  // (workgroup-uniform)
        __shared__ uint32_t s_wb[2][4];
        __shared__ uint32_t s_wbase[2];
        const bool big = on && vi.x >= walk.big, small = on && !big;
        const unsigned long long bb = __ballot(big), bs = __ballot(small);
        if (lane == 0) s_wb[0][wave] = (uint32_t)__popcll(bb), s_wb[1][wave] = (uint32_t)__popcll(bs);
        __syncthreads();
        if (tid < 2u) {  // one returning atomic per class and block
            const uint32_t n = (s_wb[tid][0] + s_wb[tid][1]) + (s_wb[tid][2] + s_wb[tid][3]);
            s_wbase[tid] = n ? atomicAdd(walk.count + tid, n) : 0u;
        }
        __syncthreads();
        if (on) {
            const uint32_t cls = big ? 0u : 1u;
            uint32_t pos = s_wbase[cls] + (uint32_t)__popcll((big ? bb : bs) & ((1ull << lane) - 1ull));
            for (uint32_t w = 0; w < wave; ++w) pos += s_wb[cls][w];
            if (!big) pos = n_chunks - 1u - pos;
            walk.items[pos] = make_uint4(c, b0 + wv + iv - sv, b1 + wi + ii - si, vi.x | ((vi.y / 6u) << 16));
            walk.li[pos] = b2 + ws + is - ss;
        }
    }
    if (bid == nb - 1 && tid == 0) {
        offsets[2 * n_chunks] = b0 + tv;
        offsets[2 * n_chunks + 1] = b1 + ti;
        offsets[2 * n_chunks + 2] = b2 + ts;
    }
}

// ---- vertex / index materials ----------------------------------------------------------------
struct VMat {
    unsigned long long ind, wgt;  // 8 material indices (byte 7 = count) and 8 weights, byte e at bits 8e
};

__device__ __forceinline__ VMat vertex_materials(const bool* has, const uint8_t* mat) {
    uint32_t ind[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wgt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t count = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (has[c]) {
            int found = -1;
#pragma unroll
            for (int e = 0; e < 7; ++e)
                if (found < 0 && (uint32_t)e < count && ind[e] == mat[c]) found = e;
            if (found < 0) {
#pragma unroll
                for (int e = 0; e < 7; ++e)
                    if ((uint32_t)e == count) {
                        ind[e] = mat[c];
                        wgt[e] = 1;
                    }
                count += 1;
            } else {
#pragma unroll
                for (int e = 0; e < 7; ++e)
                    if (e == found) wgt[e] += 1;
            }
        }
    }
    ind[7] = count;
    // sorting_network_7 (surface_nets.rs:428-446) on weights, descending, 17 compare-and-swaps
#define CSWAP(i, j)                  \
    if (wgt[i] < wgt[j]) {           \
        uint32_t t = wgt[i];         \
        wgt[i] = wgt[j];             \
        wgt[j] = t;                  \
        t = ind[i];                  \
        ind[i] = ind[j];             \
        ind[j] = t;                  \
    }
    CSWAP(0, 6) CSWAP(1, 5) CSWAP(2, 4) CSWAP(0, 3) CSWAP(1, 2) CSWAP(4, 5) CSWAP(0, 1) CSWAP(2, 3) CSWAP(4, 6) CSWAP(5, 6)
    CSWAP(1, 4) CSWAP(3, 5) CSWAP(1, 2) CSWAP(3, 4) CSWAP(5, 6) CSWAP(2, 3) CSWAP(4, 5)
#undef CSWAP
    VMat r{0ull, 0ull};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        r.ind |= (unsigned long long)(ind[e] & 0xFF) << (8 * e);
        r.wgt |= (unsigned long long)(wgt[e] & 0xFF) << (8 * e);
    }
    return r;
}

__device__ __forceinline__ uint32_t byte_at(unsigned long long v, uint32_t e) { return (uint32_t)(v >> (8 * e)) & 0xFFu; }

// calculate_index_materials_for_triangle (surface_nets.rs:559-637): out[3] = 8 bytes each (indices[4], weights[4])
__device__ __forceinline__ void index_materials(const VMat vm[3], unsigned long long out[3]) {
    const uint32_t cnt0 = byte_at(vm[0].ind, 7), cnt1 = byte_at(vm[1].ind, 7), cnt2 = byte_at(vm[2].ind, 7);
    if (cnt0 == 1 && cnt1 == 1 && cnt2 == 1) {
        const uint32_t index = byte_at(vm[0].ind, 0);
        if (byte_at(vm[1].ind, 0) == index && byte_at(vm[2].ind, 0) == index) {
            unsigned long long im = (unsigned long long)index | (1ull << 32);
            out[0] = out[1] = out[2] = im;
            return;
        }
    }
    const uint32_t cnt[3] = {cnt0, cnt1, cnt2};
    uint32_t top[4] = {0, 0, 0, 0};
    uint32_t n_top = 0;
    uint32_t off[3] = {0, 0, 0};
    bool done = false;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (!done) {
            uint32_t w[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) w[i] = byte_at(vm[i].wgt, off[i]);
            int mx = (w[0] >= w[1]) ? ((w[0] >= w[2]) ? 0 : 2) : ((w[1] >= w[2]) ? 1 : 2);
            uint32_t wmx = mx == 0 ? w[0] : (mx == 1 ? w[1] : w[2]);
            if (wmx == 0) {
                done = true;
            } else {
                uint32_t ti = mx == 0 ? byte_at(vm[0].ind, off[0]) : (mx == 1 ? byte_at(vm[1].ind, off[1]) : byte_at(vm[2].ind, off[2]));
                top[t] = ti;
                n_top = t + 1;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    for (int guard = 0; guard < 8; ++guard) {
                        if (off[i] >= cnt[i]) break;
                        uint32_t x = byte_at(vm[i].ind, off[i]);
                        bool is_top = false;
#pragma unroll
                        for (int u = 0; u < 4; ++u) is_top |= ((uint32_t)u < n_top && top[u] == x);
                        if (!is_top) break;
                        off[i] += 1;
                    }
                }
            }
        }
    }
    const unsigned long long tops = (unsigned long long)top[0] | ((unsigned long long)top[1] << 8) | ((unsigned long long)top[2] << 16) |
                                    ((unsigned long long)top[3] << 24);
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        unsigned long long wts = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if ((uint32_t)i < n_top) {
                uint32_t wv = 0;
                bool found = false;
                for (uint32_t j = 0; j < cnt[v]; ++j) {
                    if (!found && byte_at(vm[v].ind, j) == top[i]) {
                        wv = byte_at(vm[v].wgt, j);
                        found = true;
                    }
                }
                wts |= (unsigned long long)wv << (8 * i);
            }
        }
        out[v] = tops | (wts << 32);
    }
}

__device__ __forceinline__ uint32_t vertex_of(uint32_t vrow, int k) { return (vrow >> 17) + (uint32_t)__popc(vrow & ((1u << k) - 1u)); }

// a / b, correctly rounded, for operands far from the exponent limits and b != 0 (no scaling, no special cases): the hardware
// reciprocal refined once, the quotient corrected once (Markstein's sequence; the general division costs about twice as many
// instructions for its scaling and fix-ups). Used on the mesher's decoded distances and edge counts only; `ivx_selftest_mesher_division`
// compares it with the `/` operator over that whole operand set on the device (tests/test_gpu_parity.py).
__device__ __forceinline__ float div_ranged(float a, float b) {
    float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    const float q = a * y;
    const float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}

// g / |g| as the reference takes it — the correctly rounded length, then three correctly rounded divisions by it — for the gradients
// the mesher meets: the general forms cost 16 instructions for the root (rescaling of tiny inputs, class test) and 11 per division
// (scaling of both operands, fix-ups), which was a quarter of a vertex's instructions. With the squared length away from the exponent
// limits none of that can trigger: the root is the hardware estimate moved to the neighbour whose residual changes sign, a division is
// the compiler's own sequence (reciprocal refined once — shared by the three —, quotient corrected twice) without its scaling. A
// component is 0 or at least 2^-77 in magnitude (a sum of products of two centroid weights >= 2^-24 and a distance difference, all
// multiples of 2^-77), so no residual underflows; a zero numerator is +0 throughout (the weights are non-negative and one of each pair
// o, 1 - o is positive) and comes out +0 as from `/`. The caller takes this path only when every lane of the wave has
// `len2 >= 2^-100` (else the general forms: a zero gradient must come out as their NaN). Compared with `/` and sqrtf over random
// gradients by `ivx_selftest_mesher_division`.
__device__ __forceinline__ void normalize_ranged(float gx, float gy, float gz, float len2, float& nx, float& ny, float& nz) {
    const float s = __builtin_amdgcn_sqrtf(len2);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, len2), r_up = __builtin_fmaf(-s_up, s, len2);
    float b = r_dn <= 0.0f ? s_dn : s;
    b = r_up > 0.0f ? s_up : b;
    float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    auto quot = [&](float a) {
        float q = a * y;
        float r = __builtin_fmaf(-b, q, a);
        q = __builtin_fmaf(r, y, q);
        r = __builtin_fmaf(-b, q, a);
        return __builtin_fmaf(r, y, q);
    };
    nx = quot(gx), ny = quot(gy), nz = quot(gz);
}

// centroid_of_edge_intersections (surface_nets.rs:384-418) over the 12 CUBE_EDGES (661-674) of one cube. The reference adds, per crossed
// edge (c1, c2) in list order, p1 * (1 - t) + p2 * t with t = d1 / (d1 - d2) and p the corners' 0/1 coordinates. Component by component
// that term is: t along the edge's own axis (0 * (1 - t) + 1 * t), (1 - t) + t where both corners have the coordinate 1, +0 where both
// have 0 — and adding +0 leaves a non-negative sum as it is. So an edge costs one division, (1 - t) + t, and the additions of the
// components that can be non-zero, instead of six multiplications and six additions by constants the compiler may not fold
// (0 * x is not 0 for every x). Same values, same order of the additions that matter. `neg`: bit c = corner c's distance is negative.
__device__ __forceinline__ void edge_centroid(const float d[8], uint32_t neg, V3& sum, int& count) {
    sum = mk(0.0f, 0.0f, 0.0f);
    constexpr int E1[12] = {0, 0, 0, 1, 1, 2, 2, 3, 4, 4, 5, 6};
    constexpr int E2[12] = {1, 2, 4, 3, 5, 3, 6, 7, 5, 6, 7, 7};
    // Which of the twelve edges are crossed, all at once: corners c and c + a (a = 4, 2, 1: the edge's axis) differ in sign where bit c of
    // neg ^ (neg >> a) is set. Bit c1 of `cx` / `cy` / `cz` = the edge from corner c1 along x / y / z. The wave's union of the three masks
    // (DPP ORs, then four lane reads) is a SCALAR: an edge that no vertex of the wave crosses — on a locally flat surface most of the twelve,
    // e.g. every edge along the surface — is passed over by a scalar bit test and branch, without a vector instruction (the twelve per-edge
    // sign tests and ballots were a fifth of a vertex's vector instructions).
    const uint32_t cx = (neg ^ (neg >> 4)) & 0x0Fu, cy = (neg ^ (neg >> 2)) & 0x33u, cz = (neg ^ (neg >> 1)) & 0x55u;
    const uint32_t cm = cx | (cy << 8) | (cz << 16);
    count = (int)__popc(cm);
    uint32_t wm = cm;
    wm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wm, 0x111, 0xF, 0xF, true);  // row_shr:1 .. 8: lane 15 of a DPP row has the row's OR
    wm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wm, 0x112, 0xF, 0xF, true);
    wm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wm, 0x114, 0xF, 0xF, true);
    wm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wm, 0x118, 0xF, 0xF, true);
    const uint32_t any = ((uint32_t)__builtin_amdgcn_readlane((int)wm, 15) | (uint32_t)__builtin_amdgcn_readlane((int)wm, 31)) |
                         ((uint32_t)__builtin_amdgcn_readlane((int)wm, 47) | (uint32_t)__builtin_amdgcn_readlane((int)wm, 63));
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const int c1 = E1[e], c2 = E2[e];
        const int axis = c1 ^ c2;  // 4: x, 2: y, 1: z
        const uint32_t bit = 1u << (c1 + (axis == 4 ? 0 : (axis == 2 ? 8 : 16)));
        // (an edge that no vertex of the wave crosses adds +0 to every sum: skipped for the wave. Lanes of an inactive tail of the loop's last
        // round take no part in the DPP ORs — their contribution reads as 0)
        if ((any & bit) == 0u) continue;
        const bool crossed = (cm & bit) != 0u;
        const float q = div_ranged(d[c1], d[c1] - d[c2]);
        const float t = crossed ? q : 0.0f;
        const float s = crossed ? ((1.0f - q) + q) : 0.0f;
        sum.x += axis == 4 ? t : ((c1 & 4) ? s : 0.0f);
        sum.y += axis == 2 ? t : ((c1 & 2) ? s : 0.0f);
        sum.z += axis == 1 ? t : ((c1 & 1) ? s : 0.0f);
    }
}

// The loads of one chunk's padded tile, in flight: two rows per thread (324 rows: the 16 interior cells of both planes and the four k-halo
// bytes), and — in the first 27 threads — the first word of one chunk record of the 3 x 3 x 3 neighbourhood (kind, generated kind, flags,
// uniform type: what the rows need of their three chunks, the upper-layer rule of its three, the submesh of the chunk itself). A mesher
// workgroup issues them for its NEXT chunk before the quad phase of the current one (tile_issue), passes the records through LDS
// (tile_records) and looks at the rows when that chunk's turn comes (tile_finish): the tile's trip to memory — a third of a workgroup's time
// per chunk when it was taken at the start of the chunk — is then covered by the quad phase. 25 registers per thread while in flight.
struct RowData {
    uint4 s4, t4;
    uint32_t b0s, b0t, b2s, b2t;
};
struct TileLoads {
    RowData L[2];
    uint32_t rec;
};
// rim row h (0..67) of the 18 x 18 rows of a tile: the two i-faces (18 rows each), then the two j-faces (16 + 16, interleaved)
__device__ __forceinline__ void rim_row(int h, int& a, int& b) {
    a = h < 18 ? 0 : (h < 36 ? G - 1 : 1 + ((h - 36) >> 1));
    b = h < 18 ? h : (h < 36 ? h - 18 : ((h & 1) ? G - 1 : 0));
}
// Thread t takes the chunk's OWN row t — row (i, j) = (t >> 4, t & 15), tile row (i + 1, j + 1): its planes at a uniform base + 16 t, its k-halo
// bytes at a uniform base + t, its three chunk kinds the same for all 256 rows — and threads 0..67 a row of the tile's rim as well, which
// belongs to one of eight other chunk columns and takes the per-row arithmetic (half of the instructions of both steps when every row did).
__device__ __forceinline__ void tile_issue(const GridView& g, uint32_t chunk, TileLoads& T, uint32_t tid) {
    tid = opaque(tid);
    const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
    T.rec = 0u;  // (a chunk that is not there reads as Void; nothing consults it)
    if (tid < 27u) {  // first in the queue: tile_records asks for it before the quad phase's stores have retired
        const int di = (int)(tid / 9u) - 1, dj = (int)((tid / 3u) % 3u) - 1, dk = (int)(tid % 3u) - 1;
        const int ni = ci + di, nj = cj + dj, nk = ck + dk;
        if (nj >= 0 && nj < (int)g.cy && nk >= 0 && nk < (int)g.cz) {
            const ivx_chunk_info* rp = nullptr;
            if (ni >= 0 && ni < (int)g.cx) rp = g.info + (size_t)((ni * (int)g.cy + nj) * (int)g.cz + nk);
            else if (g.ghost_info[ni < 0 ? 0 : 1]) rp = g.ghost_info[ni < 0 ? 0 : 1] + (nj * (int)g.cz + nk);
            if (rp) T.rec = *reinterpret_cast<const uint32_t*>(rp);
        }
    }
    {  // the own row
        RowData& L = T.L[0];
        const size_t o = ((size_t)chunk << 12) + ((size_t)tid << 4);
        // (the k-halo bytes as the aligned WORDS around them, the byte taken in tile_finish: a byte load is a load and an extension to the
        // compiler, and it moved the extension — an operation on the loaded register — up to the load, i.e. waited for the load here)
        const uint32_t* kf = reinterpret_cast<const uint32_t*>(g.kface + (size_t)chunk * 1024) + (tid >> 2);
        const uint32_t* kf0 = ck > 0 ? kf - 256 : kf;  // (clamped to the chunk itself where there is no neighbour; not consulted then)
        const uint32_t* kf2 = ck + 1 < (int)g.cz ? kf + 256 : kf;
        L.s4 = *reinterpret_cast<const uint4*>(g.sdf + o);
        L.t4 = *reinterpret_cast<const uint4*>(g.type + o);
        L.b0s = kf0[64];
        L.b0t = kf0[192];
        L.b2s = kf2[0];
        L.b2t = kf2[128];
    }
    {  // a rim row (straight-line: a thread without one, or whose row lies outside the grid or in a ghost layer, loads the chunk's own first
       // row instead; with the loads under conditions the compiler merged the two rows' registers and waited for the first to copy it aside)
        RowData& L = T.L[1];
        int a, b;
        rim_row((int)tid, a, b);
        int gi = ci * 16 + a - 1, gj = cj * 16 + b - 1;
        const bool real = tid < 68u && row_mode(g, gi, gj) == 1u;
        gi = real ? gi : ci * 16;
        gj = real ? gj : cj * 16;
        const size_t cidx = (size_t)(((gi >> 4) * g.cy + (gj >> 4)) * g.cz) + (size_t)ck;
        const uint32_t rix = (uint32_t)(((gi & 15) << 4) | (gj & 15));
        const size_t o = (cidx << 12) + ((size_t)rix << 4);
        const uint32_t* kf = reinterpret_cast<const uint32_t*>(g.kface + cidx * 1024) + (rix >> 2);
        const uint32_t* kf0 = ck > 0 ? kf - 256 : kf;
        const uint32_t* kf2 = ck + 1 < (int)g.cz ? kf + 256 : kf;
        L.s4 = *reinterpret_cast<const uint4*>(g.sdf + o);
        L.t4 = *reinterpret_cast<const uint4*>(g.type + o);
        L.b0s = kf0[64];
        L.b0t = kf0[192];
        L.b2s = kf2[0];
        L.b2t = kf2[128];
    }
}
// the loads have to be in: the compiler's wait goes where this is called
__device__ __forceinline__ void tile_pin(TileLoads& T) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        RowData& L = T.L[r];
        asm volatile("" : "+v"(L.s4.x), "+v"(L.s4.y), "+v"(L.s4.z), "+v"(L.s4.w), "+v"(L.t4.x), "+v"(L.t4.y), "+v"(L.t4.z), "+v"(L.t4.w));
        asm volatile("" : "+v"(L.b0s), "+v"(L.b0t), "+v"(L.b2s), "+v"(L.b2t));
    }
    asm volatile("" : "+v"(T.rec));
}
// the records into LDS, ahead of the barrier that precedes tile_finish
__device__ __forceinline__ void tile_records(TileLoads& T, uint32_t* s_rec, uint32_t tid) {
    if (tid < 27u) s_rec[tid] = T.rec;
}
// one padded row into the LDS tile: `k0, k1, k2` the first record words of the row's chunks below / own / above along k
// `rix`: the row's index in its chunk (its k-halo bytes are byte rix & 3 of the words fetched)
__device__ __forceinline__ void tile_row_store(RowData& L, uint32_t rix, uint32_t w0, uint32_t w1, uint32_t w2, bool has_lo, bool has_hi, uint32_t sd[6],
                                               uint32_t ty[6]) {
    // (the loaded registers pass through an empty asm first: the compiler may otherwise move the first operation on a loaded value — a
    // mask, a shift — up to the load, a phase ago, and wait for the load there)
    asm volatile("" : "+v"(L.s4.x), "+v"(L.s4.y), "+v"(L.s4.z), "+v"(L.s4.w), "+v"(L.t4.x), "+v"(L.t4.y), "+v"(L.t4.z), "+v"(L.t4.w));
    asm volatile("" : "+v"(L.b0s), "+v"(L.b0t), "+v"(L.b2s), "+v"(L.b2t));
    const ivx_chunk_info c0 = record_of(w0), c1 = record_of(w1), c2 = record_of(w2);
    const bool d1 = c1.kind == KIND_NONUNIFORM;
    const uint32_t us = ivx_uniform_sdf(c1.kind) * 0x01010101u, ut = ivx_uniform_type(c1) * 0x01010101u;
    sd[1] = d1 ? L.s4.x : us, sd[2] = d1 ? L.s4.y : us, sd[3] = d1 ? L.s4.z : us, sd[4] = d1 ? L.s4.w : us;
    ty[1] = d1 ? L.t4.x : ut, ty[2] = d1 ? L.t4.y : ut, ty[3] = d1 ? L.t4.z : ut, ty[4] = d1 ? L.t4.w : ut;
    const uint32_t sh = 8u * (rix & 3u);
    if (has_lo) {
        const bool dense = c0.kind == KIND_NONUNIFORM;
        sd[0] = dense ? ((L.b0s >> sh) & 0xFFu) : ivx_uniform_sdf(c0.kind);
        ty[0] = dense ? ((L.b0t >> sh) & 0xFFu) : ivx_uniform_type(c0);
    }
    if (has_hi) {
        const bool dense = c2.kind == KIND_NONUNIFORM;
        sd[5] = dense ? ((L.b2s >> sh) & 0xFFu) : ivx_uniform_sdf(c2.kind);
        ty[5] = dense ? ((L.b2t >> sh) & 0xFFu) : ivx_uniform_type(c2);
    }
}
__device__ __forceinline__ void tile_row_write(int r, const uint32_t sd[6], const uint32_t ty[6], uint8_t* s_sd, uint8_t* s_ty, uint32_t* s_neg) {
    s_neg[r] = ((sd[0] >> 7) & 1u) | (neg16(sd + 1) << 1) | (((sd[5] >> 7) & 1u) << 17);
    uint8_t* ds = s_sd + r * RS;
    uint8_t* dt = s_ty + r * RS;
    ds[3] = (uint8_t)sd[0];
    dt[3] = (uint8_t)ty[0];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        reinterpret_cast<uint32_t*>(ds + 4)[q] = sd[1 + q];
        reinterpret_cast<uint32_t*>(dt + 4)[q] = ty[1 + q];
    }
    ds[20] = (uint8_t)sd[5];
    dt[20] = (uint8_t)ty[5];
}
__device__ __forceinline__ void tile_finish(const GridView& g, uint32_t chunk, TileLoads& T, const uint32_t* s_rec, uint8_t* s_sd, uint8_t* s_ty, uint32_t* s_neg,
                                            uint32_t tid, int* upper) {
    tid = opaque(tid);
    const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
    const bool has_lo = ck > 0, has_hi = ck + 1 < (int)g.cz;
    {  // the own row (always in the grid; chunk column (0, 0) of the neighbourhood: records 12, 13, 14)
        uint32_t sd[6], ty[6];
        sd[0] = sd[5] = 0x7Fu;
        ty[0] = ty[5] = 0xFFu;
        tile_row_store(T.L[0], tid, s_rec[12], s_rec[13], s_rec[14], has_lo, has_hi, sd, ty);
        tile_row_write((int)(((tid >> 4) + 1u) * G + (tid & 15u) + 1u), sd, ty, s_sd, s_ty, s_neg);
    }
    if (tid < 68u) {  // a rim row
        int a, b;
        rim_row((int)tid, a, b);
        const int gi = ci * 16 + a - 1, gj = cj * 16 + b - 1;
        uint32_t sd[6], ty[6];
        sd[0] = sd[5] = 0x7Fu;
        ty[0] = ty[5] = 0xFFu;
        sd[1] = sd[2] = sd[3] = sd[4] = 0x7F7F7F7Fu;
        ty[1] = ty[2] = ty[3] = ty[4] = 0xFFFFFFFFu;
        const uint32_t mode = row_mode(g, gi, gj);
        if (mode == 2u) {
            fetch_row_serial(g, gi, gj, ck, sd, ty);
        } else if (mode == 1u) {
            // the row's chunk column in the neighbourhood: (gi >> 4) - ci and (gj >> 4) - cj are -1, 0 or 1
            const int col = (((gi >> 4) - ci + 1) * 3 + ((gj >> 4) - cj + 1)) * 3;
            tile_row_store(T.L[1], (uint32_t)(((gi & 15) << 4) | (gj & 15)), s_rec[col], s_rec[col + 1], s_rec[col + 2], has_lo, has_hi, sd, ty);
        }
        tile_row_write(a * G + b, sd, ty, s_sd, s_ty, s_neg);
    }
    // the upper layer of cubes belongs to the upper neighbour chunk when that chunk is non-uniform (surface_nets.rs:252-261); neighbourhood
    // entries (di, dj, dk) = (1,0,0), (0,1,0), (0,0,1); a chunk that is not there reads as Void
    upper[0] = upper[1] = upper[2] = G - 1;
    if ((s_rec[22] & 0xFFu) == KIND_NONUNIFORM) upper[0] -= 1;
    if ((s_rec[16] & 0xFFu) == KIND_NONUNIFORM) upper[1] -= 1;
    if ((s_rec[14] & 0xFFu) == KIND_NONUNIFORM) upper[2] -= 1;
}

// Workgroups of a mesher launch: as many as stay resident (four per CU: 128 VGPRs, < 40 KB of LDS), each walking its share of the list —
// only a workgroup that goes on to another chunk can fetch that chunk's tile ahead.
static inline uint32_t ivx_emit_grid(const ivx_grid* g, uint32_t n_entries) {
    const uint32_t resident = (uint32_t)g->ctx->n_cu * 4u;
    return n_entries < resident ? (n_entries ? n_entries : 1u) : resident;
}
static inline uint32_t ivx_emit_general_grid(const ivx_grid* g, uint32_t n_entries) {
    const uint32_t cap = (uint32_t)g->ctx->n_cu * 2u;
    return n_entries < cap ? (n_entries ? n_entries : 1u) : cap;
}

// The mesh buffers' stores go out as NONTEMPORAL ones: written once, read by nobody in this launch (the quad phase takes its vertices from
// LDS), 1.6 GB of them per all-surface step that would otherwise push the tiles' lines — which neighbouring chunks are about to read — out of
// the L2. Measured side by side on one box (tools/ab_bench.sh): mesher 0.470-0.480 -> 0.453-0.457 ms on the all-surface grid, 57.5 -> 55.9 us
// on the asteroid. (The same run: staging a wave's indices through LDS so that they too leave as consecutive 16-byte stores changed nothing;
// with NO mesh store at all the launch takes 0.32 ms — the stores cost 0.13 ms although the memory side is at 0.55 of its peak: loads queue
// behind a CU's own stores.)
#define IVX_MESH_ST(p, v) __builtin_nontemporal_store((v), (p))
// The mesher's main pass. One workgroup (256 threads, four per CU: <= 128 VGPRs by amdgpu_waves_per_eu(4) on the kernels) walks its share of
// the chunks that have a mesh, as a two-stage pipeline over chunks — the tile of the walk's NEXT chunk is fetched while the current chunk is
// meshed from the tile in LDS:
//   0. next tile out  the loads of the next chunk's padded tile are issued (TileLoads: 25 registers per thread)
//   1. order          cube rows in scan order, one ordered prefix for vertices and quads  (tile signs -> s_vrow, s_qrow, s_surf)
//   2. vertices       one thread per vertex: position, normal, the one material; the vertex lists its quads (tile -> buffers, s_vpos, s_vsm, s_quad)
//   4. quads          one thread per quad: diagonal, winding, indices, index materials    (reads nothing of the tile but its sign rows)
//   5. next tile in   the rows that arrived meanwhile go into the LDS tile
// The tile's trip to memory, a third of a workgroup's time per chunk when it was taken at the start of the chunk, is covered by phases 1-4.
// This pass does the common case only — every vertex with one material around it, every quad with one material at its four corners, at most
// VPC vertices — and hands any other chunk on (`hard`) to role_sn_emit_general, which meshes that chunk again in full: kept apart so that
// the general paths' registers are not live beside the loads in flight (with them in the same kernel the compiler sent the loaded rows to
// scratch the moment they arrived, i.e. waited for them where they were issued).
template <bool SLOTS>
__device__ __forceinline__ void role_sn_emit(uint32_t bid, uint32_t nb, SnParams p, float* __restrict__ positions, float* __restrict__ normals,
                                                 uint32_t* __restrict__ indices, unsigned long long* __restrict__ imats, ivx_submesh* __restrict__ submeshes,
                                                 const uint32_t* __restrict__ emit_count, const uint4* __restrict__ emit_items, uint32_t vcap, uint32_t icap,
                                                 uint32_t scap, const uint32_t* __restrict__ slots, uint32_t* __restrict__ hard_count,
                                                 uint32_t* __restrict__ hard_list, uint32_t* __restrict__ cursor, SnWalk walk = SnWalk{nullptr, nullptr, nullptr, 0u},
                                                 uint32_t walk_len = 0u) {
    __shared__ uint16_t s_quad[3 * VPC];  // the chunk's quads in emission order: cube id | axis << 13
    __shared__ __attribute__((aligned(16))) uint8_t s_sd[TILE_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t s_ty[TILE_BYTES];
    __shared__ uint32_t s_neg2[2][NROWS];  // the tile's sign rows, double-buffered (see the barriers of a round)
    __shared__ uint32_t s_vrow[NCROWS];   // per cube row: first vertex << 17 | which of its 17 cubes have a vertex (cube -> vertex: vertex_of)
    __shared__ uint16_t s_qrow[NCROWS];   // per cube row: its first quad
    __shared__ uint2 s_qbits[NCROWS];     // per cube row: which cubes emit their X / Y / Z quad — x = qx | qy << 17, y = qy >> 15 | qz << 2 (17 bits each)
    __shared__ uint32_t s_nq;             // the chunk's quads
    __shared__ uint16_t s_surf[VPC];      // vertex -> cube id (cube row * 17 + k)
    __shared__ float s_vpos[3][VPC];      // the chunk's vertex positions and materials for the quad phase
    __shared__ uint8_t s_vsm[VPC];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_hard;  // this chunk needs the general pass
    __shared__ uint32_t s_rec[27];  // first words of the chunk records of the tile's 3 x 3 x 3 neighbourhood
    __shared__ uint32_t s_ticket[5];  // the list entry this workgroup takes after the next one: index, record
    __shared__ float s_rcount[13];    // 1 / n for a cube's 1..12 crossed edges: a read instead of a division per vertex
    const GridView& g = p.g;
    const uint32_t tid = threadIdx.x;
    if (tid >= 1u && tid < 13u) s_rcount[tid] = 1.0f / (float)tid;  // (visible after the first round's first barrier)
    const uint32_t n_emit = emit_count[0];  // = number of submeshes (k_sn_scan's third total)
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    // The list (submesh order: entry li is submesh li) is handed out by counters: a workgroup starts on entry `bid` and draws every further one.
    // Equal shares did not end together — identical chunks took 9 to 21 us (10th to 90th percentile) depending on what else the CU was doing,
    // and the launch waited 150 us of its 600 for the slowest shares. Eight counters, each on a cache line of its own, one per residue of the
    // block index mod 8 (blocks of one residue share an XCD): counter s hands out the entries nb + 8 t + s, so a thousand workgroups drawing
    // at once queue on eight words, not one (one word serves ~90 draws per microsecond). The draw runs two rounds ahead (the entry after the
    // next, whose tile is already being fetched), by thread 0 only: ticket at the top of a round, the entry's record when the ticket has come
    // back, both through LDS at the round's last barrier — no other wave ever waits for either.
    // (`walk`: position t of the walk is the entry walk.li[w(t)] with the record walk.items[w(t)], w(t) = t among the entries listed from the
    // front, else counted from the back of the arrays — role_sn_scan; without it position t is entry t)
    const uint32_t walk_front = walk.items ? walk.count[0] : 0u;
    uint32_t li = NONE, li_next = bid, li_prev = NONE;
    if (li_next >= n_emit) return;
    uint4 item = make_uint4(0u, 0u, 0u, 0u), item_next;
    if (walk.items) {
        const uint32_t w = bid < walk_front ? bid : walk_len - 1u - (bid - walk_front);
        item_next = walk.items[w];
        li_next = walk.li[w];
    } else {
        item_next = emit_items[li_next];
    }
    int upper[3] = {G - 1, G - 1, G - 1};  // of the tile in LDS
    uint32_t info_w = 0u;
    uint32_t buf = 0u;         // s_neg2[buf]: the sign rows of the tile in LDS (the next tile's go to the other half while this chunk's quads still read these)
    bool from_ticket = false;  // (li_next, item_next) wait in s_ticket
    constexpr uint32_t TK = 192u;  // the thread that draws the tickets: lane 0 of the last wave, the one with the fewest vertex rounds
    // Barriers of a round: (1) top — the tile, the drawn entry and the previous chunk's `hard` verdict are published; (2) inside the ordered
    // prefix; (3) the vertex order is built; (4) the vertices are in LDS, and so are the next tile's records. Nothing after the quads: a wave
    // goes from its quads straight to putting the next tile into LDS (the tile's planes are dead after the vertex phase, its sign rows are
    // double-buffered) and on to the next round's first barrier — seven barriers per chunk came down to four.
    for (;;) {
    const bool have = li != NONE;
    const uint32_t* s_neg = s_neg2[buf];
    const uint32_t chunk = item.x, voff = item.y, ioff = item.z;
    const uint32_t icount = (item.w >> 16) * 6u;
    const uint32_t slot = have ? (SLOTS ? slots[li] : li) : 0u;  // (incremental remesh: the submesh manager's slot of the chunk)
    // The output buffers keep the capacity of earlier steps; a mesh that outgrew them is re-emitted after the host has grown the
    // buffers (ivx_voxel_step_collect) — nothing is ever written past the end: such a chunk is passed over.
    const bool fits = have && !((size_t)voff + (item.w & 0xFFFFu) > vcap || (size_t)ioff + icount > icap || slot >= scap);
    const bool large = fits && (item.w & 0xFFFFu) > VPC;
    const uint32_t vcount = (fits && !large) ? (item.w & 0xFFFFu) : 0u;
    const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
    if (have) IVX_T(g, li, 0);
    if (tid == 0 && fits) {
        const uint32_t cflags = (info_w >> 16) & 0xFFu;
        ivx_submesh sm;
        sm.chunk_indices[0] = (uint32_t)ci + p.x_off;
        sm.chunk_indices[1] = (uint32_t)cj;
        sm.chunk_indices[2] = (uint32_t)ck;
        sm.index_offset = ioff;
        sm.index_count = icount;
        // bits: X_DN 0, Y_DN 1, Z_DN 2, X_UP 3, Y_UP 4, Z_UP 5 (mesh.rs:611-635)
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int c = 0; c < 2; ++c)
                    sm.is_obscured_from_direction[a][b][c] =
                        (((cflags >> (3 * a)) & 1u) && ((cflags >> (3 * b + 1)) & 1u) && ((cflags >> (3 * c + 2)) & 1u)) ? 1u : 0u;
        sm.vertex_offset = voff;
        sm.vertex_count = item.w & 0xFFFFu;
        sm.reserved = 0;
        submeshes[slot] = sm;
    }
    __syncthreads();  // (1) the chunk's tile is in LDS (phase 5 of the previous round); every wave is through the previous chunk's quads
    if (have) IVX_T(g, li, 1);
    if (from_ticket) {
        li_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket[0]);
        item_next = make_uint4((uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket[1]), (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket[2]),
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket[3]), (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket[4]));
    }
    const bool have_next = li_next < n_emit;
    if (tid == 0) {  // the previous chunk's verdict (its quad phase is over everywhere), then this chunk's starting value
        if (li_prev != NONE && s_hard) hard_list[atomicAdd(hard_count, 1u)] = li_prev;
        s_hard = large ? 1u : 0u;
    }

    uint32_t ticket = 0u;
    if (tid == TK && have_next) ticket = nb + 8u * atomicAdd(cursor + 32u * (bid & 7u), 1u) + (bid & 7u);
    // ---- 0. the next chunk's tile: its loads travel while this chunk's vertex order is built
    TileLoads T;
    if (have_next) tile_issue(g, item_next.x, T, tid);
    uint4 drawn = make_uint4(0u, 0u, 0u, 0u);
    uint32_t drawn_li = 0u;
    // (straight-line from here to the vertices, also in a round without any — the first one, a chunk passed over —, where the order is built
    // from whatever the tile's LDS holds and nobody looks at it: with the phases under a condition the two paths kept the loads in flight in
    // different registers and met behind a full drain)
    // ---- 1. vertex and quad order: cubes in (i,j,k) scan order (surface_nets.rs:158-185) = cube rows in order, bits ascending; a cube's
    // quads X, Y, Z (surface_nets.rs:263-301). Thread t owns cube rows 2t and 2t+1 so that thread order = row order; ONE ordered prefix
    // carries both counts (vertices in the low half, quads in the high half: at most 4913 and 14 739).
    {
        uint32_t vb[2] = {0, 0}, nq[2] = {0, 0};
        const int r0 = 2 * (int)opaque(tid);
        for (int q = 0; q < 2; ++q) {
            const int cr = r0 + q;
            if (cr < NCROWS) {
                uint32_t qx, qy, qz;
                cube_row_bits(s_neg, cr / 17, cr % 17, upper, vb[q], qx, qy, qz);
                nq[q] = __popc(qx) + __popc(qy) + __popc(qz);
                s_qbits[cr] = make_uint2(qx | (qy << 17), (qy >> 15) | (qz << 2));  // (for the vertices' quad lists: visible behind the prefix's barrier)
            }
        }
        uint32_t total;
        const uint32_t pre = block_prefix<false>((__popc(vb[0]) + __popc(vb[1])) | ((nq[0] + nq[1]) << 16), s_wsum, tid, total);  // (2)
        uint32_t base = pre & 0xFFFFu, qb = pre >> 16;
        if (tid == 0) s_nq = total >> 16;
        for (int q = 0; q < 2; ++q) {
            const int cr = r0 + q;
            if (cr < NCROWS) {
                s_vrow[cr] = (base << 17) | vb[q];
                s_qrow[cr] = (uint16_t)qb;
                qb += nq[q];
                uint32_t m = vb[q];
                while (m) {
                    const int k = __ffs(m) - 1;
                    m &= m - 1;
                    if (base < VPC) s_surf[base] = (uint16_t)(cr * 17 + k);
                    base += 1;
                }
            }
        }
    }
    // The next tile's loads are waited for HERE, before this round's first store. The counter of outstanding vector-memory operations is
    // in order: behind the vertex and quad phases' stores — whose number the compiler cannot know, the loops' trip counts being data — the
    // wait for these loads was a wait for every store of the round to be acknowledged, once per chunk, in every wave. Here only the previous
    // round's quad stores are older, and the order phase has covered most of their trip and of the loads'.
    if (have_next) tile_pin(T);
    if (have) IVX_T(g, li, 6);  // (order built and the next tile's loads in, this wave)
    __syncthreads();  // (3)
    if (have) IVX_T(g, li, 2);  // vertex order built
    // the entry behind the ticket (the atomic is older than the tile's loads: it has returned): one wave's uniform load, waited for before that
    // wave's first store for the same reason
    if ((tid >> 6) == (TK >> 6)) {
        const uint32_t tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
        if (have_next && tk < n_emit) {
            if (walk.items) {
                const uint32_t w = tk < walk_front ? tk : walk_len - 1u - (tk - walk_front);
                drawn = walk.items[w];
                drawn_li = walk.li[w];
            } else {
                drawn = emit_items[tk];
                drawn_li = tk;
            }
            asm volatile("" : "+v"(drawn.x), "+v"(drawn.y), "+v"(drawn.z), "+v"(drawn.w), "+v"(drawn_li));
        } else {
            drawn_li = tk;  // (beyond the list: the walk ends there)
        }
    }

    // mesh.rs:559-577
    const float chunk_extent = p.extent * 16.0f;
    const V3 pos_offset = mk((float)(ci + (int)p.x_off) * chunk_extent - 0.5f * p.extent, (float)cj * chunk_extent - 0.5f * p.extent,
                             (float)ck * chunk_extent - 0.5f * p.extent);

    // ---- 2. vertices: one thread per vertex ------------------------------------------------------
    // (Rounds of 64 vertices per wave with EVERY lane at work — a lane beyond the chunk's last vertex redoes that vertex and stores nothing —,
    // so that the wave-wide edge mask of edge_centroid can be taken with DPP and lane reads; a wave with no vertex in a round skips it: the
    // bound is wave-uniform.)
    for (uint32_t v0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid & ~63u)); v0 < vcount; v0 += 256) {
        const uint32_t v_raw = v0 + (opaque(tid) & 63u);
        const bool v_live = v_raw < vcount;
        const uint32_t v = v_live ? v_raw : vcount - 1u;
        const int cid = s_surf[v];
        const int cr = cid / 17, k = cid - cr * 17, i = cr / 17, j = cr - i * 17;
        const int t0 = tix(i, j, k);
        const int co[8] = {0, 1, RS, RS + 1, G * RS, G * RS + 1, G * RS + RS, G * RS + RS + 1};
        float d[8];
        uint32_t neg = 0u;
        uint8_t mats[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int8_t e = (int8_t)s_sd[t0 + co[c]];
            d[c] = decode(e);
            neg |= (e < 0 ? 1u : 0u) << c;
            mats[c] = s_ty[t0 + co[c]];
        }
        int count;
        V3 sum;
        edge_centroid(d, neg, sum, count);
        const float rc = s_rcount[count];
        const V3 centroid = scale(sum, rc);
        // trilinear gradient (object/sdf.rs:603-633)
        const V3 d00 = sub(mk(d[4], d[2], d[1]), mk(d[0], d[0], d[0]));
        const V3 d01 = sub(mk(d[5], d[6], d[3]), mk(d[1], d[4], d[2]));
        const V3 d10 = sub(mk(d[6], d[3], d[5]), mk(d[2], d[1], d[4]));
        const V3 d11 = sub(mk(d[7], d[7], d[7]), mk(d[3], d[5], d[6]));
        const V3 o = centroid;
        const V3 r = sub(mk(1.0f, 1.0f, 1.0f), o);
        const V3 r_yzx = mk(r.y, r.z, r.x), r_zxy = mk(r.z, r.x, r.y), o_yzx = mk(o.y, o.z, o.x), o_zxy = mk(o.z, o.x, o.y);
        const V3 grad = add(add(add(mul(mul(r_yzx, r_zxy), d00), mul(mul(r_yzx, o_zxy), d01)), mul(mul(o_yzx, r_zxy), d10)),
                            mul(mul(o_yzx, o_zxy), d11));
        const float gl2 = (grad.x * grad.x + grad.y * grad.y) + grad.z * grad.z;
        V3 normal;
        if (__builtin_amdgcn_ballot_w64(!(gl2 >= 0x1p-100f)) == 0ull) {
            normalize_ranged(grad.x, grad.y, grad.z, gl2, normal.x, normal.y, normal.z);
        } else {
            const float gl = sqrtf(gl2);
            normal = mk(grad.x / gl, grad.y / gl, grad.z / gl);
        }
        const V3 position = add(scale(add(centroid, mk((float)i, (float)j, (float)k)), p.extent), pos_offset);
        // the one material of the cube's negative corners; a vertex with several hands the chunk to the general pass
        uint32_t m0 = 0xFFFFFFFFu;
        bool single = true;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if ((neg >> c) & 1u) {
                if (m0 == 0xFFFFFFFFu) m0 = mats[c];
                single = single && mats[c] == m0;
            }
        // the vertex's quads: at the row's first quad + the quads of the row's cubes below this one (the row's three masks from the order phase)
        const uint2 qb = s_qbits[cr];
        const uint32_t below = (1u << k) - 1u;
        const uint32_t qslot0 = (uint32_t)s_qrow[cr] + __popc(qb.x & (below | (below << 17))) + __popc(qb.y & ((below >> 15) | (below << 2)));
        if (v_live) {
            const size_t gv = (size_t)voff + v;
            IVX_MESH_ST(positions + 3 * gv + 0, position.x);
            IVX_MESH_ST(positions + 3 * gv + 1, position.y);
            IVX_MESH_ST(positions + 3 * gv + 2, position.z);
            IVX_MESH_ST(normals + 3 * gv + 0, normal.x);
            IVX_MESH_ST(normals + 3 * gv + 1, normal.y);
            IVX_MESH_ST(normals + 3 * gv + 2, normal.z);
            s_vpos[0][v] = position.x;
            s_vpos[1][v] = position.y;
            s_vpos[2][v] = position.z;
            s_vsm[v] = (uint8_t)m0;
            if (!single) s_hard = 1u;
            uint32_t qslot = qslot0;
            if ((qb.x >> k) & 1u) s_quad[qslot++] = (uint16_t)cid;                                            // X
            if (k < 15 ? ((qb.x >> (17 + k)) & 1u) : ((qb.y >> (k - 15)) & 1u)) s_quad[qslot++] = (uint16_t)(cid | (1 << 13));  // Y
            if ((qb.y >> (2 + k)) & 1u) s_quad[qslot] = (uint16_t)(cid | (2 << 13));                              // Z
        }
    }
    if (have) IVX_T(g, li, 7);  // (wave 0's vertices written)
    if (have_next) tile_records(T, s_rec, tid);  // (published by the barrier below: every wave needs them right after its quads)
    __syncthreads();  // (4)
    if (have) IVX_T(g, li, 3);  // vertices written


    // ---- 4. quads, one THREAD per quad of the list the vertices left (a vertex emits up to three quads: walking them inside its thread
    // made every pass three quads long). No barrier inside or behind: the waves drift apart and cover each other's LDS and store latencies.
    // Rounds of 64 quads per wave with every lane at work, like the vertices': the index materials of a wave's 64 quads — 384 entries of
    // 8 bytes, the same entry six times per quad — go out as three stores of 1 KiB of CONSECUTIVE bytes each (lane l writes entries 2x and
    // 2x + 1, x = l, l + 64, l + 128, of quad x / 3, whose material it fetches from that quad's lane) instead of three stores that each cover
    // a third of every cache line of the 3 KiB: a third of the write requests for two thirds of a quad's bytes.
    if (vcount && !s_hard) {
        const uint32_t n_quads = s_nq;
        for (uint32_t q0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid & ~63u)); q0 < n_quads; q0 += 256) {
            const uint32_t lane = opaque(tid) & 63u;
            const uint32_t q_raw = q0 + lane;
            const bool q_live = q_raw < n_quads;
            const uint32_t q = q_live ? q_raw : n_quads - 1u;
            const uint32_t qd = s_quad[q];
            const int axis = (int)(qd >> 13);
            const int qcid = (int)(qd & 0x1FFFu);
            const int cr = qcid / 17;
            const int k = qcid - cr * 17, i = cr / 17, j = cr - i * 17;
            // neighbouring cubes across the two other axes: (axis_b, axis_c) = (Y,Z), (Z,X), (X,Y)
            const int ab = axis == 0 ? 17 : (axis == 1 ? 1 : 289);
            const int ac = axis == 0 ? 1 : (axis == 1 ? 289 : 17);
            const bool n1 = ((s_neg[i * G + j] >> k) & 1u) != 0u;  // the cube's first corner is inside
            const bool negative_face = !n1;  // (false,true) => negative face (surface_nets.rs:348-352)
            // cube -> vertex: the row's first vertex plus the vertices below k in the row (ab, ac step the row by 0, 1 or 17 and k by 0 or 1)
            const int rb = ab == 1 ? 0 : (ab == 17 ? 1 : 17), rc = ac == 1 ? 0 : (ac == 17 ? 1 : 17);
            const int kb = ab == 1 ? 1 : 0, kc = ac == 1 ? 1 : 0;
            const uint32_t v1 = vertex_of(s_vrow[cr], k), v2 = vertex_of(s_vrow[cr - rb], k - kb), v3 = vertex_of(s_vrow[cr - rc], k - kc),
                           v4 = vertex_of(s_vrow[cr - rb - rc], k - kb - kc);
            const V3 q1 = mk(s_vpos[0][v1], s_vpos[1][v1], s_vpos[2][v1]), q2 = mk(s_vpos[0][v2], s_vpos[1][v2], s_vpos[2][v2]);
            const V3 q3 = mk(s_vpos[0][v3], s_vpos[1][v3], s_vpos[2][v3]), q4 = mk(s_vpos[0][v4], s_vpos[1][v4], s_vpos[2][v4]);
            const uint32_t b1 = s_vsm[v1], b2 = s_vsm[v2], b3 = s_vsm[v3], b4 = s_vsm[v4];
            uint32_t quad[6];
            // the shorter diagonal: |q1 - q4| < |q2 - q3| on the rounded lengths (surface_nets.rs:360-381). Where the SQUARED lengths already
            // say "not shorter" the (correctly rounded, hence monotonic) roots cannot say otherwise; only a wave with a quad whose squared
            // lengths differ the other way takes the roots — none on a flat piece of surface, where the diagonals are equal.
            const V3 e14 = sub(q1, q4), e23 = sub(q2, q3);
            const float sq14 = (e14.x * e14.x + e14.y * e14.y) + e14.z * e14.z, sq23 = (e23.x * e23.x + e23.y * e23.y) + e23.z * e23.z;
            bool first = sq14 < sq23;
            if (__builtin_amdgcn_ballot_w64(first) != 0ull) first = sqrtf(sq14) < sqrtf(sq23);
            if (first) {
                if (negative_face) { quad[0] = v1; quad[1] = v4; quad[2] = v2; quad[3] = v1; quad[4] = v3; quad[5] = v4; }
                else { quad[0] = v1; quad[1] = v2; quad[2] = v4; quad[3] = v1; quad[4] = v4; quad[5] = v3; }
            } else if (negative_face) { quad[0] = v2; quad[1] = v3; quad[2] = v4; quad[3] = v2; quad[4] = v1; quad[5] = v3; }
            else { quad[0] = v2; quad[1] = v4; quad[2] = v3; quad[3] = v2; quad[4] = v3; quad[5] = v1; }
            if (q_live) {
                const size_t io = (size_t)ioff + (size_t)q * 6;
#pragma unroll
                for (int t = 0; t < 6; ++t) IVX_MESH_ST(indices + io + t, voff + quad[t]);
                if (!(b1 == b2 && b1 == b3 && b1 == b4)) s_hard = 1u;  // (corners of different materials: the general pass redoes the chunk)
            }
            // calculate_index_materials_for_triangle's first case (surface_nets.rs:559-637): one entry, weight 1
            const uint32_t rem3 = 3u * min(n_quads - q0, 64u);  // pairs of entries of this wave's quads
            uint4* imw = reinterpret_cast<uint4*>(imats + (size_t)ioff + (size_t)q0 * 6);
#pragma unroll
            for (int jx = 0; jx < 3; ++jx) {
                const uint32_t x = lane + 64u * (uint32_t)jx;
                const uint32_t src = (x * 171u) >> 9;  // x / 3 for x < 192
                const uint32_t bm = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)b1);
                if (x < rem3) {
                    uint32_t* w_ = reinterpret_cast<uint32_t*>(imw + x);
                    IVX_MESH_ST(w_, bm), IVX_MESH_ST(w_ + 1, 1u), IVX_MESH_ST(w_ + 2, bm), IVX_MESH_ST(w_ + 3, 1u);
                }
            }
        }
    }
    if (tid == TK && have_next) {
        s_ticket[0] = ticket < n_emit ? drawn_li : NONE;
        s_ticket[1] = drawn.x, s_ticket[2] = drawn.y, s_ticket[3] = drawn.z, s_ticket[4] = drawn.w;
    }
    if (have) IVX_T(g, li, 4);  // (wave 0's) quads written

    // ---- 5. the next chunk's tile into LDS
    if (have_next) {
        tile_finish(g, item_next.x, T, s_rec, s_sd, s_ty, s_neg2[buf ^ 1u], tid, upper);
        info_w = s_rec[13];
    }
    if (have) IVX_T(g, li, 5);
    if (!have_next) break;
    li_prev = li;
    li = li_next;
    item = item_next;
    from_ticket = true;
    buf ^= 1u;
    }
    // the last chunk's verdict
    __syncthreads();
    if (tid == 0 && li != NONE && s_hard) hard_list[atomicAdd(hard_count, 1u)] = li;
}

// The general mesher: one workgroup per listed chunk, every case of the reference's vertex and index materials (vertices with up to seven
// materials around them, the sorting network, calculate_index_materials_for_triangle's ranking), chunks of any vertex count. It runs after
// role_sn_emit over the chunks that one handed on (`hard` = count, then submesh-order list entries) and writes the whole chunk again.
template <bool SLOTS>
__device__ __forceinline__ void role_sn_emit_general(uint32_t bid, uint32_t nb, SnParams p, float* __restrict__ positions, float* __restrict__ normals,
                                                 uint32_t* __restrict__ indices, unsigned long long* __restrict__ imats,
                                                 uint4* __restrict__ vmats, ivx_submesh* __restrict__ submeshes, const uint32_t* __restrict__ emit_count,
                                                 const uint4* __restrict__ emit_items, uint32_t vcap, uint32_t icap, uint32_t scap,
                                                 const uint32_t* __restrict__ slots, const uint32_t* __restrict__ hard_count,
                                                 const uint32_t* __restrict__ hard_list) {
    __shared__ uint16_t s_quad[768];  // quads of the current batch of 256 vertices: cube id | axis << 13
    __shared__ __attribute__((aligned(16))) uint8_t s_sd[TILE_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t s_ty[TILE_BYTES];
    __shared__ uint32_t s_neg[NROWS];
    __shared__ uint32_t s_vrow[NCROWS];   // per cube row: first vertex << 17 | which of its 17 cubes have a vertex (cube -> vertex: vertex_of)
    __shared__ uint16_t s_surf[NCUBES];   // vertex -> cube id (cube row * 17 + k)
    // the chunk's vertices for the quad phase, when there are at most VPC of them (else that phase reads them back from memory):
    // positions, and the one material around the vertex (0xFF: several) — with these 11 KB the workgroup takes 40 656 bytes of LDS,
    // four to a CU like its 128 registers
    __shared__ float s_vpos[3][VPC];
    __shared__ uint8_t s_vsm[VPC];
    __shared__ uint32_t s_wsum[4];
    const GridView& g = p.g;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_emit = emit_count[0];  // = number of submeshes (k_sn_scan's third total)
    const uint32_t n_hard = hard_count[0] < n_emit ? hard_count[0] : n_emit;
    for (uint32_t hi = bid; hi < n_hard; hi += nb) {
    const uint32_t li = hard_list[hi];
    __syncthreads();
    if (li >= n_emit) continue;
    const uint4 item = emit_items[li];  // the list is in submesh order: entry li is submesh li
    const uint32_t chunk = item.x, voff = item.y, ioff = item.z;
    const uint32_t vcount = item.w & 0xFFFFu, icount = (item.w >> 16) * 6u;
    // the output buffers keep the capacity of earlier steps; a mesh that outgrew them is re-emitted after the host has
    // grown the buffers (ivx_voxel_step_collect) — nothing is ever written past the end
    const uint32_t slot = SLOTS ? slots[li] : li;  // (incremental remesh: the submesh manager's slot of the chunk)
    if ((size_t)voff + vcount > vcap || (size_t)ioff + icount > icap || slot >= scap) continue;
    const ivx_chunk_info info = g.info[chunk];
    const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
    UpperLoads ul;
    upper_issue(g, ci, cj, ck, ul);
    load_tile(g, ci, cj, ck, s_sd, s_ty, s_neg, tid);
    int upper[3];
    upper_finish(g, ci, cj, ck, ul, upper);

    if (tid == 0) {
        ivx_submesh sm;
        sm.chunk_indices[0] = (uint32_t)ci + p.x_off;
        sm.chunk_indices[1] = (uint32_t)cj;
        sm.chunk_indices[2] = (uint32_t)ck;
        sm.index_offset = ioff;
        sm.index_count = icount;
        // bits: X_DN 0, Y_DN 1, Z_DN 2, X_UP 3, Y_UP 4, Z_UP 5 (mesh.rs:611-635)
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int c = 0; c < 2; ++c)
                    sm.is_obscured_from_direction[a][b][c] =
                        (((info.flags >> (3 * a)) & 1u) && ((info.flags >> (3 * b + 1)) & 1u) && ((info.flags >> (3 * c + 2)) & 1u)) ? 1u : 0u;
        sm.vertex_offset = voff;
        sm.vertex_count = vcount;
        sm.reserved = 0;
        submeshes[slot] = sm;
    }
    __syncthreads();

    // ---- vertex order: cubes in (i,j,k) scan order (surface_nets.rs:158-185) = cube rows in order, bits ascending.
    // Thread t owns cube rows 2t and 2t+1 so that thread order = row order for the ordered prefix.
    {
        uint32_t vb[2] = {0, 0};
        const int r0 = 2 * (int)tid;
        for (int q = 0; q < 2; ++q) {
            const int cr = r0 + q;
            if (cr < NCROWS) {
                uint32_t qx, qy, qz;
                cube_row_bits(s_neg, cr / 17, cr % 17, upper, vb[q], qx, qy, qz);
            }
        }
        uint32_t total;
        uint32_t base = block_prefix(__popc(vb[0]) + __popc(vb[1]), s_wsum, tid, total);
        for (int q = 0; q < 2; ++q) {
            const int cr = r0 + q;
            if (cr < NCROWS) {
                s_vrow[cr] = (base << 17) | vb[q];
                uint32_t m = vb[q];
                while (m) {
                    const int k = __ffs(m) - 1;
                    m &= m - 1;
                    s_surf[base] = (uint16_t)(cr * 17 + k);
                    base += 1;
                }
            }
        }
    }
    __syncthreads();

    // mesh.rs:559-577
    const float chunk_extent = p.extent * 16.0f;
    const V3 pos_offset = mk((float)(ci + (int)p.x_off) * chunk_extent - 0.5f * p.extent, (float)cj * chunk_extent - 0.5f * p.extent,
                             (float)ck * chunk_extent - 0.5f * p.extent);

    // ---- phase A: one thread per vertex (dense) -------------------------------------------------
    for (uint32_t v = tid; v < vcount; v += 256) {
        const int cid = s_surf[v];
        const int cr = cid / 17, k = cid - cr * 17, i = cr / 17, j = cr - i * 17;
        const int t0 = tix(i, j, k);
        const int co[8] = {0, 1, RS, RS + 1, G * RS, G * RS + 1, G * RS + RS, G * RS + RS + 1};
        float d[8];
        bool has[8];
        uint8_t mats[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int8_t e = (int8_t)s_sd[t0 + co[c]];
            d[c] = decode(e);
            has[c] = e < 0;
            mats[c] = s_ty[t0 + co[c]];
        }
        // centroid of edge intersections (surface_nets.rs:384-418)
        const int E1[12] = {0, 0, 0, 1, 1, 2, 2, 3, 4, 4, 5, 6};
        const int E2[12] = {1, 2, 4, 3, 5, 3, 6, 7, 5, 6, 7, 7};
        int count = 0;
        V3 sum = mk(0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const int c1 = E1[e], c2 = E2[e];
            const float d1 = d[c1], d2 = d[c2];
            if (sneg(d1) != sneg(d2)) {
                count += 1;
                const float interp1 = d1 / (d1 - d2);
                const float interp2 = 1.0f - interp1;
                const V3 p1 = mk((float)((c1 >> 2) & 1), (float)((c1 >> 1) & 1), (float)(c1 & 1));
                const V3 p2 = mk((float)((c2 >> 2) & 1), (float)((c2 >> 1) & 1), (float)(c2 & 1));
                sum = add(sum, add(scale(p1, interp2), scale(p2, interp1)));
            }
        }
        const float rc = 1.0f / (float)count;
        const V3 centroid = scale(sum, rc);
        // trilinear gradient (object/sdf.rs:603-633)
        const V3 d00 = sub(mk(d[4], d[2], d[1]), mk(d[0], d[0], d[0]));
        const V3 d01 = sub(mk(d[5], d[6], d[3]), mk(d[1], d[4], d[2]));
        const V3 d10 = sub(mk(d[6], d[3], d[5]), mk(d[2], d[1], d[4]));
        const V3 d11 = sub(mk(d[7], d[7], d[7]), mk(d[3], d[5], d[6]));
        const V3 o = centroid;
        const V3 r = sub(mk(1.0f, 1.0f, 1.0f), o);
        const V3 r_yzx = mk(r.y, r.z, r.x), r_zxy = mk(r.z, r.x, r.y), o_yzx = mk(o.y, o.z, o.x), o_zxy = mk(o.z, o.x, o.y);
        const V3 grad = add(add(add(mul(mul(r_yzx, r_zxy), d00), mul(mul(r_yzx, o_zxy), d01)), mul(mul(o_yzx, r_zxy), d10)),
                            mul(mul(o_yzx, o_zxy), d11));
        const float gl = len3(grad);
        const V3 normal = mk(grad.x / gl, grad.y / gl, grad.z / gl);
        const V3 position = add(scale(add(centroid, mk((float)i, (float)j, (float)k)), p.extent), pos_offset);
        // Most bodies are of one material around most vertices: when every lane's non-empty corners agree on theirs (a wave vote), the
        // table is that one entry — what the general construction (first-seen table, counts, the 17-swap sorting network: ~400 predicated
        // instructions) returns for this input
        VMat vm;
        {
            uint32_t m0 = 0xFFFFFFFFu, n_in = 0u;
            bool single = true;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (has[c]) {
                    if (m0 == 0xFFFFFFFFu) m0 = mats[c];
                    single = single && mats[c] == m0;
                    n_in += 1u;
                }
            if (__all(single ? 1 : 0)) {
                vm.ind = (unsigned long long)(m0 & 0xFFu) | (1ull << 56);
                vm.wgt = (unsigned long long)n_in;
            } else {
                vm = vertex_materials(has, mats);
            }
            if (v < VPC) {
                s_vpos[0][v] = position.x;
                s_vpos[1][v] = position.y;
                s_vpos[2][v] = position.z;
                s_vsm[v] = single ? (uint8_t)m0 : (uint8_t)0xFFu;  // (0xFF is the type of empty voxels, never a vertex material)
            }
        }
        const size_t gv = (size_t)voff + v;
        positions[3 * gv + 0] = position.x;
        positions[3 * gv + 1] = position.y;
        positions[3 * gv + 2] = position.z;
        normals[3 * gv + 0] = normal.x;
        normals[3 * gv + 1] = normal.y;
        normals[3 * gv + 2] = normal.z;
        vmats[gv] = make_uint4((uint32_t)vm.ind, (uint32_t)(vm.ind >> 32), (uint32_t)vm.wgt, (uint32_t)(vm.wgt >> 32));
    }
    __threadfence_block();
    __syncthreads();

    // ---- phase B: quads in surface-point order, X then Y then Z edge (surface_nets.rs:263-301). Per batch of 256 vertices
    // the quads are first listed in that order (ordered prefix over the vertices' quad counts), then handled one THREAD per
    // quad: a vertex emits up to three quads, and walking them inside its thread made every pass three quads long.
    uint32_t qbase = 0;
    const bool cached = vcount <= VPC;  // (the same for the whole workgroup)
    for (uint32_t v0 = 0; v0 < vcount; v0 += 256) {
        const uint32_t v = v0 + tid;
        uint32_t qm = 0;
        uint32_t cid = 0;
        if (v < vcount) {
            cid = s_surf[v];
            const int cr = cid / 17;
            const int k = cid - cr * 17, i = cr / 17, j = cr - i * 17;
            uint32_t vb, qx, qy, qz;
            cube_row_bits(s_neg, i, j, upper, vb, qx, qy, qz);
            qm = ((qx >> k) & 1u) | (((qy >> k) & 1u) << 1) | (((qz >> k) & 1u) << 2);
        }
        uint32_t total;
        uint32_t slot = block_prefix(__popc(qm), s_wsum, tid, total);  // (ends with a barrier: the previous batch's list is consumed)
#pragma unroll
        for (uint32_t axis = 0; axis < 3; ++axis)
            if ((qm >> axis) & 1u) s_quad[slot++] = (uint16_t)(cid | (axis << 13));
        __syncthreads();
        for (uint32_t q = tid; q < total; q += 256) {
            const uint32_t qd = s_quad[q];
            const int axis = (int)(qd >> 13);
            const int qcid = (int)(qd & 0x1FFFu);
            const int cr = qcid / 17;
            const int k = qcid - cr * 17, i = cr / 17, j = cr - i * 17;
            // neighbouring cubes across the two other axes: (axis_b, axis_c) = (Y,Z), (Z,X), (X,Y)
            const int ab = axis == 0 ? 17 : (axis == 1 ? 1 : 289);
            const int ac = axis == 0 ? 1 : (axis == 1 ? 289 : 17);
            const bool n1 = (int8_t)s_sd[tix(i, j, k)] < 0;
            const bool negative_face = !n1;  // (false,true) => negative face (surface_nets.rs:348-352)
            // cube -> vertex: the row's first vertex plus the vertices below k in the row (ab, ac step the row by 0, 1 or 17 and k by 0 or 1)
            const int rb = ab == 1 ? 0 : (ab == 17 ? 1 : 17), rc = ac == 1 ? 0 : (ac == 17 ? 1 : 17);
            const int kb = ab == 1 ? 1 : 0, kc = ac == 1 ? 1 : 0;
            const uint32_t v1 = vertex_of(s_vrow[cr], k), v2 = vertex_of(s_vrow[cr - rb], k - kb), v3 = vertex_of(s_vrow[cr - rc], k - kc),
                           v4 = vertex_of(s_vrow[cr - rb - rc], k - kb - kc);
            // The four corners and their vertex materials: from LDS when the chunk's vertices fit there; a wave whose quads each have one
            // material at all four corners (nearly every wave of nearly every body) then needs nothing from memory. Otherwise positions
            // and materials are read back from the buffers, all loads issued together (the materials travel with the positions: fetched
            // per triangle corner after the diagonal is chosen they were a second dependent round trip to memory).
            V3 q1, q2, q3, q4;
            uint4 vm1 = make_uint4(0, 0, 0, 0), vm2 = vm1, vm3 = vm1, vm4 = vm1;
            uint32_t one_mat = 0xFFu;
            bool fast = false;
            if (cached) {
                q1 = mk(s_vpos[0][v1], s_vpos[1][v1], s_vpos[2][v1]), q2 = mk(s_vpos[0][v2], s_vpos[1][v2], s_vpos[2][v2]);
                q3 = mk(s_vpos[0][v3], s_vpos[1][v3], s_vpos[2][v3]), q4 = mk(s_vpos[0][v4], s_vpos[1][v4], s_vpos[2][v4]);
                const uint32_t b1 = s_vsm[v1], b2 = s_vsm[v2], b3 = s_vsm[v3], b4 = s_vsm[v4];
                one_mat = b1;
                fast = __all((b1 != 0xFFu && b1 == b2 && b1 == b3 && b1 == b4) ? 1 : 0) != 0;
            } else {
                const float* P = positions + 3 * (size_t)voff;
                q1 = mk(P[3 * v1], P[3 * v1 + 1], P[3 * v1 + 2]), q2 = mk(P[3 * v2], P[3 * v2 + 1], P[3 * v2 + 2]);
                q3 = mk(P[3 * v3], P[3 * v3 + 1], P[3 * v3 + 2]), q4 = mk(P[3 * v4], P[3 * v4 + 1], P[3 * v4 + 2]);
            }
            if (!fast) {
                vm1 = vmats[(size_t)voff + v1], vm2 = vmats[(size_t)voff + v2], vm3 = vmats[(size_t)voff + v3], vm4 = vmats[(size_t)voff + v4];
                asm volatile("" : "+v"(vm1.x), "+v"(vm1.y), "+v"(vm1.z), "+v"(vm1.w), "+v"(vm2.x), "+v"(vm2.y), "+v"(vm2.z), "+v"(vm2.w));
                asm volatile("" : "+v"(vm3.x), "+v"(vm3.y), "+v"(vm3.z), "+v"(vm3.w), "+v"(vm4.x), "+v"(vm4.y), "+v"(vm4.z), "+v"(vm4.w));
            }
            uint32_t quad[6];
            if (len3(sub(q1, q4)) < len3(sub(q2, q3))) {
                if (negative_face) { quad[0] = v1; quad[1] = v4; quad[2] = v2; quad[3] = v1; quad[4] = v3; quad[5] = v4; }
                else { quad[0] = v1; quad[1] = v2; quad[2] = v4; quad[3] = v1; quad[4] = v4; quad[5] = v3; }
            } else if (negative_face) { quad[0] = v2; quad[1] = v3; quad[2] = v4; quad[3] = v2; quad[4] = v1; quad[5] = v3; }
            else { quad[0] = v2; quad[1] = v4; quad[2] = v3; quad[3] = v2; quad[4] = v3; quad[5] = v1; }
            const size_t io = (size_t)ioff + (size_t)(qbase + q) * 6;
#pragma unroll
            for (int t = 0; t < 6; ++t) indices[io + t] = voff + quad[t];
            if (fast) {  // calculate_index_materials_for_triangle's first case (surface_nets.rs:559-637): one entry, weight 1
                const unsigned long long im = (unsigned long long)one_mat | (1ull << 32);
#pragma unroll
                for (int t = 0; t < 6; ++t) imats[io + t] = im;
                continue;
            }
#pragma unroll
            for (int tri = 0; tri < 2; ++tri) {
                VMat vm[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint32_t vq = quad[3 * tri + c];
                    const uint4 raw = vq == v1 ? vm1 : (vq == v2 ? vm2 : (vq == v3 ? vm3 : vm4));
                    vm[c].ind = (unsigned long long)raw.x | ((unsigned long long)raw.y << 32);
                    vm[c].wgt = (unsigned long long)raw.z | ((unsigned long long)raw.w << 32);
                }
                unsigned long long im[3];
                index_materials(vm, im);
                imats[io + 3 * tri + 0] = im[0];
                imats[io + 3 * tri + 1] = im[1];
                imats[io + 3 * tri + 2] = im[2];
            }
        }
        qbase += total;
    }
    }
}


}  // namespace sn
}  // namespace ivx_roles
