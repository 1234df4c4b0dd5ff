// Contact generation between two voxel objects (second part): collision probes picked from the mesh, then each
// object's probes sampled against the other object's signed distance field.
//
// Reference (engine/crates/impact_voxel/src):
//   VoxelObjectCollisionProbes::recompute_for_all_chunks      collidable.rs:361-392, 451-523
//   add_points_for_vertices_in_blocks                          collidable.rs:614-731 (block index helpers 733-789)
//   for_each_mutual_voxel_object_contact                       collidable.rs:859-1049
//   determine_sdf_value_and_normal_at_point_if_intersecting    collidable.rs:1288-1440
//   evaluate_sdf_from_corner_samples / compute_sdf_gradient_from_corner_samples   object/sdf.rs:579-633
//   ContactID::from_two_u64_and_n_indices                      impact_physics/src/constraint/contact.rs:180-199
//
// Probes. One workgroup per chunk submesh. The reference accumulates two curvature samples per triangle corner into the corner's
// vertex while walking the triangles in index order; f32 addition does not commute with reordering, so every vertex gathers its
// corners (count, scan, fill in LDS; the corner lists themselves live in a global scratch the size of the index buffer), replays
// them in ascending corner order, and only then competes for its block with a 64-bit LDS min over (curvature, vertex index) —
// which reproduces "first vertex with the strictly smallest curvature wins". Probes leave the chunk in block order, chunks in
// submesh order (count per chunk, scan, gather).
//
// Contacts. One thread per probe: box test, transform into the other object's normalized space, eight corner samples of its SDF
// (Void chunks read +2.54, Uniform ones -2.56, as VoxelObject::voxel does), trilinear value and gradient. Contacts keep probe
// order: count per workgroup of 256 probes, one scan over the workgroups of both passes (A's probes first, then B's), ordered
// emit.
#include "ivx_internal.hpp"

namespace {

constexpr uint32_t PROBE_MAXV = 4928u;  // a chunk's Surface Nets vertices: one per cube of the 17^3 the chunk owns, at most
constexpr uint32_t PROBE_SMALLV = 1024u;

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 cmul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
struct Q4 {
    float x, y, z, w;
};
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {  // glam Quat::mul_vec3a
    const V3 b = mk(q.x, q.y, q.z);
    const float b2 = dot(b, b);
    return (v * (q.w * q.w - b2) + b * (dot(v, b) * 2.0f)) + cross(b, v) * (q.w * 2.0f);
}
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ unsigned long long splitmix(unsigned long long state) {  // impact_math/src/random/splitmix.rs:4-10
    state += 0x9E3779B97F4A7C15ull;
    unsigned long long z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// `f32 as usize`, as far as a voxel index can matter: negative and NaN give 0, anything beyond the grid stays beyond it
__device__ __forceinline__ uint32_t as_index(float f) {
    const uint32_t u = (uint32_t)f;  // v_cvt_u32_f32 saturates
    return u > 0x7FFFFFF0u ? 0x7FFFFFF0u : u;
}

// ---- probes ----------------------------------------------------------------------------------------------------------------------
// exclusive scan of `mine` over the 256 threads of the workgroup; total in *total
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t mine, uint32_t* s_w, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o) incl += t;
    }
    __syncthreads();  // (s_w may still be read from a previous call)
    if (lane == 63u) s_w[wave] = incl;
    __syncthreads();
    const uint32_t w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    const uint32_t wbase = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
    *total = (w0 + w1) + (w2 + w3);
    return wbase + incl - mine;
}

// MAXV = the most vertices a workgroup of this variant takes (its LDS footprint: 12 B per vertex). Chunk meshes have a few hundred
// vertices, the worst case is 4913: the launcher runs a small variant (1024 vertices, 12 KB: many workgroups per CU) for the chunks that fit
// it and the full-size one for the rest; a workgroup whose chunk belongs to the other variant leaves at once.
struct ProbeSelectArgs {
    const ivx_submesh* submeshes;
    const float* pos;
    const float* nrm;
    const uint32_t* idx;
    uint32_t* corner_list;
    uint32_t* sel;
    uint32_t* counts;
    uint32_t* counts_host;  // (optional) the counts a second time: host-mapped memory, for a host that waits for many objects at once
    uint32_t* err;          // set to 1 by plain stores (may be host-mapped)
    const uint32_t* slots;
    uint32_t log2_bs;
    float inv_extent;
};
template <uint32_t MAXV, bool SMALL>
__device__ __forceinline__ void probe_select_body(const ProbeSelectArgs& a, uint32_t bid, uint32_t) {
    const ivx_submesh* __restrict__ submeshes = a.submeshes;
    const float* __restrict__ pos = a.pos;
    const float* __restrict__ nrm = a.nrm;
    const uint32_t* __restrict__ idx = a.idx;
    uint32_t* __restrict__ corner_list = a.corner_list;
    uint32_t* __restrict__ sel = a.sel;
    uint32_t* __restrict__ counts = a.counts;
    const uint32_t* __restrict__ slots = a.slots;
    const uint32_t log2_bs = a.log2_bs;
    const float inv_extent = a.inv_extent;
    __shared__ unsigned long long s_mem[MAXV];  // two u32 per vertex first, then the block table (MAXV x 8 B at most)
    __shared__ float s_curv[MAXV];
    __shared__ uint32_t s_w[4];
    uint32_t* s_start = reinterpret_cast<uint32_t*>(s_mem);
    uint32_t* s_fill = s_start + MAXV;
    unsigned long long* s_best = s_mem;
    const uint32_t tid = threadIdx.x, s = bid;  // s = record: outputs are indexed by it
    const ivx_submesh sm = submeshes[slots ? slots[s] : s];  // (incremental sync: the listed submesh slots)
    const uint32_t ioff = sm.index_offset, icnt = sm.index_count, voff = sm.vertex_offset, vcnt = sm.vertex_count;
    const uint32_t log2_cb = 4u - log2_bs, n_blocks = 1u << (3u * log2_cb);
    if (SMALL ? vcnt > PROBE_SMALLV : (vcnt <= PROBE_SMALLV && n_blocks <= PROBE_SMALLV)) return;  // the other variant's chunk
    if (vcnt > MAXV || n_blocks > MAXV) {  // (cannot happen for a Surface Nets chunk; never index LDS out of bounds)
        if (tid == 0) {
            counts[s] = 0;
            if (a.counts_host) a.counts_host[s] = 0;
            *reinterpret_cast<volatile uint32_t*>(a.err) = 1u;
        }
        return;
    }
    for (uint32_t v = tid; v < vcnt; v += 256u) s_fill[v] = 0;
    __syncthreads();
    for (uint32_t c = tid; c < icnt; c += 256u) atomicAdd(&s_fill[idx[ioff + c] - voff], 1u);
    __syncthreads();
    {  // exclusive scan of the corner counts: thread t owns vertices [t * per, (t + 1) * per)
        const uint32_t per = (vcnt + 255u) / 256u, v0 = tid * per, v1 = min(vcnt, v0 + per);
        uint32_t mine = 0;
        for (uint32_t v = v0; v < v1; ++v) mine += s_fill[v];
        uint32_t total;
        uint32_t run = wg_exclusive_scan(mine, s_w, &total);
        for (uint32_t v = v0; v < v1; ++v) {
            s_start[v] = run;
            run += s_fill[v];
        }
    }
    __syncthreads();
    for (uint32_t v = tid; v < vcnt; v += 256u) s_fill[v] = 0;
    __syncthreads();
    for (uint32_t c = tid; c < icnt; c += 256u) {
        const uint32_t v = idx[ioff + c] - voff;
        corner_list[ioff + s_start[v] + atomicAdd(&s_fill[v], 1u)] = c;
    }
    __syncthreads();
    // replay every vertex's corners in triangle order (collidable.rs:650-680)
    for (uint32_t v = tid; v < vcnt; v += 256u) {
        const uint32_t n = s_fill[v];
        const uint32_t* list = corner_list + ioff + s_start[v];
        const V3 nv = ld3(nrm + 3 * (size_t)(voff + v));
        float sum = 0.0f, count = 0.0f;
        long long last = -1;
        for (uint32_t r = 0; r < n; ++r) {
            uint32_t c = 0xFFFFFFFFu;
            for (uint32_t e = 0; e < n; ++e) {
                const uint32_t x = list[e];
                if ((long long)x > last && x < c) c = x;
            }
            last = (long long)c;
            const uint32_t tri = c / 3u, k = c - 3u * tri;
            const V3 p0 = ld3(pos + 3 * (size_t)idx[ioff + 3u * tri]), p1 = ld3(pos + 3 * (size_t)idx[ioff + 3u * tri + 1u]),
                     p2 = ld3(pos + 3 * (size_t)idx[ioff + 3u * tri + 2u]);
            const V3 e01 = p1 - p0, e12 = p2 - p1, e20 = p0 - p2;
            const float sample = k == 0 ? dot(nv, e01) - dot(nv, e20) : (k == 1 ? dot(nv, e12) - dot(nv, e01) : dot(nv, e20) - dot(nv, e12));
            sum += sample;
            count += 2.0f;
        }
        s_curv[v] = count == 0.0f ? __uint_as_float(0x7FC00000u) : sum / count;  // unconnected vertices never compete (NaN)
    }
    __syncthreads();
    for (uint32_t b = tid; b < n_blocks; b += 256u) s_best[b] = ~0ull;
    __syncthreads();
    const float clo[3] = {(float)(sm.chunk_indices[0] * 16u), (float)(sm.chunk_indices[1] * 16u), (float)(sm.chunk_indices[2] * 16u)};
    for (uint32_t v = tid; v < vcnt; v += 256u) {
        float c = s_curv[v];
        if (!(c < __uint_as_float(0x7F800000u))) continue;  // `curvature < min_curvature` can never hold for NaN or +inf
        if (c == 0.0f) c = 0.0f;                             // (-0.0 and +0.0 tie)
        const V3 p = ld3(pos + 3 * (size_t)(voff + v));
        const float pn[3] = {p.x * inv_extent, p.y * inv_extent, p.z * inv_extent};
        uint32_t bi[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float q = pn[d] > clo[d] ? pn[d] : clo[d];  // clamp to the chunk (max_with, then min_with)
            q = q < clo[d] + 16.0f ? q : clo[d] + 16.0f;
            bi[d] = (as_index(q) & 15u) >> log2_bs;
        }
        const uint32_t block = (bi[0] << (2u * log2_cb)) + (bi[1] << log2_cb) + bi[2];
        const uint32_t u = __float_as_uint(c);
        const uint32_t ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        atomicMin(&s_best[block], ((unsigned long long)ord << 32) | (unsigned long long)v);
    }
    __syncthreads();
    {  // chosen vertices leave in block order
        const uint32_t per = (n_blocks + 255u) / 256u, b0 = tid * per, b1 = min(n_blocks, b0 + per);
        uint32_t mine = 0;
        for (uint32_t b = b0; b < b1; ++b) mine += s_best[b] != ~0ull ? 1u : 0u;
        uint32_t total;
        uint32_t run = wg_exclusive_scan(mine, s_w, &total);
        for (uint32_t b = b0; b < b1; ++b) {
            const unsigned long long k = s_best[b];
            if (k != ~0ull) sel[(size_t)s * n_blocks + run++] = voff + (uint32_t)(k & 0xFFFFFFFFull);
        }
        if (tid == 0) {
            counts[s] = total;
            if (a.counts_host) a.counts_host[s] = total;
        }
    }
}
template <uint32_t MAXV, bool SMALL>
__global__ __launch_bounds__(256) void k_probe_select(ProbeSelectArgs a) {
    probe_select_body<MAXV, SMALL>(a, blockIdx.x, gridDim.x);
}
__device__ __forceinline__ void probe_select_small_body(const ProbeSelectArgs& a, uint32_t bid, uint32_t nb) { probe_select_body<PROBE_SMALLV, true>(a, bid, nb); }
__device__ __forceinline__ void probe_select_full_body(const ProbeSelectArgs& a, uint32_t bid, uint32_t nb) { probe_select_body<PROBE_MAXV, false>(a, bid, nb); }
IVX_MANY_TWIN(k_probe_select_small_many, ProbeSelectArgs, probe_select_small_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_probe_select_small, k_probe_select_small_many, ProbeSelectArgs, 256)
IVX_MANY_TWIN(k_probe_select_full_many, ProbeSelectArgs, probe_select_full_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_probe_select_full, k_probe_select_full_many, ProbeSelectArgs, 256)

// exclusive scan over n counts (one workgroup); offsets[n] = total
struct ScanCountsArgs {
    const uint32_t* counts;
    uint32_t* offsets;
    uint32_t* total_out;  // (optional) the total a second time: host-mapped memory, for a host that waits for many scans at once
    uint32_t n, pad;
};
__device__ __forceinline__ void scan_counts_body(const ScanCountsArgs& a, uint32_t, uint32_t) {
    const uint32_t n = a.n;
    const uint32_t* __restrict__ counts = a.counts;
    uint32_t* __restrict__ offsets = a.offsets;
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n; b0 += 256u) {
        const uint32_t b = b0 + tid;
        const uint32_t v = b < n ? counts[b] : 0u;
        uint32_t total;
        const uint32_t ex = wg_exclusive_scan(v, s_w, &total);
        if (b < n) offsets[b] = s_carry + ex;
        __syncthreads();
        if (tid == 0) s_carry += total;
        __syncthreads();
    }
    if (tid == 0) {
        offsets[n] = s_carry;
        if (a.total_out) *a.total_out = s_carry;
    }
}
__global__ __launch_bounds__(256) void k_scan_counts(ScanCountsArgs a) { scan_counts_body(a, 0u, 1u); }
IVX_MANY_TWIN(k_scan_counts_many, ScanCountsArgs, scan_counts_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_scan_counts, k_scan_counts_many, ScanCountsArgs, 256)

struct ProbeGatherArgs {
    const ivx_submesh* submeshes;
    const float* pos;
    const uint32_t* sel;
    const uint32_t* counts;
    const uint32_t* offsets;
    float* points;
    uint32_t* probe_chunk;
    uint32_t* entries;
    const uint32_t* slots;
    uint32_t n_blocks, pad;
};
__device__ __forceinline__ void probe_gather_body(const ProbeGatherArgs& a, uint32_t bid, uint32_t) {
    const ivx_submesh* __restrict__ submeshes = a.submeshes;
    const float* __restrict__ pos = a.pos;
    const uint32_t* __restrict__ sel = a.sel;
    const uint32_t* __restrict__ counts = a.counts;
    const uint32_t* __restrict__ offsets = a.offsets;
    float* __restrict__ points = a.points;
    uint32_t* __restrict__ probe_chunk = a.probe_chunk;
    uint32_t* __restrict__ entries = a.entries;
    const uint32_t* __restrict__ slots = a.slots;
    const uint32_t n_blocks = a.n_blocks;
    const uint32_t s = bid, n = counts[s], off = offsets[s];
    const ivx_submesh sm = submeshes[slots ? slots[s] : s];
    const uint32_t packed = sm.chunk_indices[0] | (sm.chunk_indices[1] << 10) | (sm.chunk_indices[2] << 20);
    for (uint32_t r = threadIdx.x; r < n; r += 64u) {
        const uint32_t v = sel[(size_t)s * n_blocks + r];
        points[3 * (size_t)(off + r)] = pos[3 * (size_t)v];
        points[3 * (size_t)(off + r) + 1] = pos[3 * (size_t)v + 1];
        points[3 * (size_t)(off + r) + 2] = pos[3 * (size_t)v + 2];
        probe_chunk[off + r] = packed;
    }
    if (threadIdx.x == 0 && entries) {
        uint32_t* e = entries + 5 * (size_t)s;
        e[0] = sm.chunk_indices[0], e[1] = sm.chunk_indices[1], e[2] = sm.chunk_indices[2], e[3] = off, e[4] = off + n;
    }
}

__global__ __launch_bounds__(64) void k_probe_gather(ProbeGatherArgs a) { probe_gather_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_probe_gather_many, ProbeGatherArgs, probe_gather_body, __launch_bounds__(64))
IVX_MANY_LAUNCHER(many_probe_gather, k_probe_gather_many, ProbeGatherArgs, 64)
static_assert(sizeof(ProbeSelectArgs) % 8 == 0 && sizeof(ProbeGatherArgs) % 8 == 0, "argument blocks travel as 8-byte words");

// ---- mutual contacts -------------------------------------------------------------------------------------------------------------
struct MutParams {
    // the sampled object
    const int8_t* sdf;
    const ivx_chunk_info* info;
    uint32_t cy, cz;
    uint32_t dims[3];  // its grid in voxels
    float center[3];   // its centre of mass, normalized
    float inv_s, ext_s;
    Q4 q_s, q_s_inv;
    float t_s[3];
    // the probing object
    Q4 q_p_inv;
    float t_p[3];
    float inv_p;
    float box_lo[3], box_hi[3];
    uint32_t clo[3], chi[3];  // chunk ranges that can touch the other object (chi exclusive)
    uint32_t n_probes;
    uint32_t negate;  // the contact normal is B's outward normal: minus A's when A is the one being sampled
    unsigned long long id_ab;
    uint32_t body_a, body_b;
    float restitution, static_friction, dynamic_friction;
};

struct MutHit {
    V3 pos, nrm;
    float depth;
    uint32_t ijk[3];
};

__device__ __forceinline__ float voxel_sd(const MutParams& p, uint32_t i, uint32_t j, uint32_t k) {  // VoxelObject::voxel(i, j, k).signed_distance().to_f32()
    const uint32_t chunk = ((i >> 4) * p.cy + (j >> 4)) * p.cz + (k >> 4);
    const uint32_t kind = p.info[chunk].kind;
    int sd;
    if (kind == KIND_NONUNIFORM) sd = (int)p.sdf[(size_t)chunk * IVX_CHUNK_VOXELS + (((i & 15u) << 8) | ((j & 15u) << 4) | (k & 15u))];
    else sd = kind == KIND_UNIFORM ? -128 : 127;
    return (float)sd * 0.02f;
}

__device__ __forceinline__ bool deep_inside(const MutParams& p, V3 np, float& sd, V3& n) {  // collidable.rs:1424-1440
    sd = -128.0f * 0.02f;
    const V3 d = np - mk(p.center[0], p.center[1], p.center[2]);
    const float n2 = dot(d, d);
    if (!(n2 > 1e-8f * 1e-8f)) return false;
    const float len = sqrtf(n2);
    n = mk(d.x / len, d.y / len, d.z / len);
    return true;
}

__device__ __forceinline__ bool sample_if_intersecting(const MutParams& p, V3 np, float& sd, V3& n) {
    const V3 lp = np - mk(0.5f, 0.5f, 0.5f);
    if ((__float_as_uint(lp.x) | __float_as_uint(lp.y) | __float_as_uint(lp.z)) & 0x80000000u) return false;  // has_negative_component: sign bits
    const uint32_t li = as_index(lp.x), lj = as_index(lp.y), lk = as_index(lp.z);
    if ((li + 1u >= p.dims[0]) | (lj + 1u >= p.dims[1]) | (lk + 1u >= p.dims[2])) return false;
    const uint32_t ci = as_index(np.x), cj = as_index(np.y), ck = as_index(np.z);
    const uint32_t chunk = ((ci >> 4) * p.cy + (cj >> 4)) * p.cz + (ck >> 4);
    const uint32_t kind = p.info[chunk].kind;
    if (kind == KIND_UNIFORM) return deep_inside(p, np, sd, n);
    if (kind == KIND_VOID) return false;
    const float containing = (float)(int)p.sdf[(size_t)chunk * IVX_CHUNK_VOXELS + (((ci & 15u) << 8) | ((cj & 15u) << 4) | (ck & 15u))] * 0.02f;
    if (containing > 0.5f * 1.7320508075688772f) return false;
    float d[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) d[c] = voxel_sd(p, li + ((c >> 2) & 1), lj + ((c >> 1) & 1), lk + (c & 1));
    const V3 off = mk(lp.x - floorf(lp.x), lp.y - floorf(lp.y), lp.z - floorf(lp.z));
    const V3 rev = mk(1.0f - off.x, 1.0f - off.y, 1.0f - off.z);
    {
        const float d00 = d[0] * rev.x + d[4] * off.x, d01 = d[1] * rev.x + d[5] * off.x, d10 = d[2] * rev.x + d[6] * off.x, d11 = d[3] * rev.x + d[7] * off.x;
        const float d0 = d00 * rev.y + d10 * off.y, d1 = d01 * rev.y + d11 * off.y;
        sd = d0 * rev.z + d1 * off.z;
    }
    if (sd > 0.0f) return false;
    if (fabsf(sd - (-128.0f * 0.02f)) < 1e-3f) return deep_inside(p, np, sd, n);
    const V3 e00 = mk(d[4] - d[0], d[2] - d[0], d[1] - d[0]), e01 = mk(d[5] - d[1], d[6] - d[4], d[3] - d[2]), e10 = mk(d[6] - d[2], d[3] - d[1], d[5] - d[4]),
             e11 = mk(d[7] - d[3], d[7] - d[5], d[7] - d[6]);
    const V3 r_yzx = mk(rev.y, rev.z, rev.x), r_zxy = mk(rev.z, rev.x, rev.y), o_yzx = mk(off.y, off.z, off.x), o_zxy = mk(off.z, off.x, off.y);
    const V3 g = ((cmul(cmul(r_yzx, r_zxy), e00) + cmul(cmul(r_yzx, o_zxy), e01)) + cmul(cmul(o_yzx, r_zxy), e10)) + cmul(cmul(o_yzx, o_zxy), e11);
    const float g2 = dot(g, g);
    if (!(g2 > 1e-8f * 1e-8f)) return false;
    const float len = sqrtf(g2);
    n = mk(g.x / len, g.y / len, g.z / len);
    return true;
}

template <bool EMIT>
__device__ __forceinline__ bool probe_contact(const MutParams& p, const float* points, const uint32_t* probe_chunk, uint32_t k, MutHit* hit) {
    if (k >= p.n_probes) return false;
    const uint32_t pc = probe_chunk[k];
    if (pc == 0xFFFFFFFFu) return false;  // a freed range of the point buffer (incremental probe sync)
    const uint32_t c3[3] = {pc & 1023u, (pc >> 10) & 1023u, pc >> 20};
    if (c3[0] < p.clo[0] || c3[0] >= p.chi[0] || c3[1] < p.clo[1] || c3[1] >= p.chi[1] || c3[2] < p.clo[2] || c3[2] >= p.chi[2]) return false;
    const V3 pp = ld3(points + 3 * (size_t)k);
    const V3 dl = pp - mk(p.box_lo[0], p.box_lo[1], p.box_lo[2]), dh = mk(p.box_hi[0], p.box_hi[1], p.box_hi[2]) - pp;
    if ((__float_as_uint(dl.x) | __float_as_uint(dl.y) | __float_as_uint(dl.z) | __float_as_uint(dh.x) | __float_as_uint(dh.y) | __float_as_uint(dh.z)) &
        0x80000000u)
        return false;  // AxisAlignedBox::contains_point: sign-bit test
    const V3 point = qrot(p.q_p_inv, pp - mk(p.t_p[0], p.t_p[1], p.t_p[2]));
    const V3 np = (qrot(p.q_s, point) + mk(p.t_s[0], p.t_s[1], p.t_s[2])) * p.inv_s;
    float sd;
    V3 n;
    if (!sample_if_intersecting(p, np, sd, n)) return false;
    if (EMIT) {
        V3 sn = qrot(p.q_s_inv, n);
        if (p.negate) sn = mk(-sn.x, -sn.y, -sn.z);
        hit->pos = point;
        hit->nrm = sn;
        hit->depth = -sd * p.ext_s;
#ifdef IVX_MUTATION_CHECK  // tools/mutation_check.sh: one ulp off in code the single-pair and the batched call share — only the oracle can tell
        hit->depth = __uint_as_float(__float_as_uint(hit->depth) + 1u);
#endif
        const V3 q = pp * p.inv_p;
        hit->ijk[0] = as_index(q.x), hit->ijk[1] = as_index(q.y), hit->ijk[2] = as_index(q.z);
    }
    return true;
}

struct MutCountArgs {
    MutParams p;
    const float* points;
    const uint32_t* probe_chunk;
    uint32_t* counts;
};
struct MutEmitArgs {
    MutParams p;
    const float* points;
    const uint32_t* probe_chunk;
    const uint32_t* offsets;
    ivx_contact* out;
    uint32_t cap, pad;
};
__device__ __forceinline__ void mut_count_body(const MutCountArgs& a, uint32_t bid, uint32_t) {
    __shared__ uint32_t s_w[4];
    const bool hit = probe_contact<false>(a.p, a.points, a.probe_chunk, bid * 256u + threadIdx.x, nullptr);
    const uint32_t n = (uint32_t)__popcll(__ballot(hit));
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) a.counts[bid] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
__global__ __launch_bounds__(256) void k_mut_count(MutCountArgs a) { mut_count_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_mut_count_many, MutCountArgs, mut_count_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_mut_count, k_mut_count_many, MutCountArgs, 256)

__device__ __forceinline__ void mut_emit_body(const MutEmitArgs& a, uint32_t bid, uint32_t) {
    const MutParams& p = a.p;
    const uint32_t* __restrict__ offsets = a.offsets;
    const uint32_t cap = a.cap;
    ivx_contact* __restrict__ out = a.out;
    __shared__ uint32_t s_w[4];
    MutHit h;
    const bool hit = probe_contact<true>(p, a.points, a.probe_chunk, bid * 256u + threadIdx.x, &h);
    const unsigned long long ballot = __ballot(hit);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    if (!hit) return;
    const uint32_t w0 = s_w[0], w1 = s_w[1], w2 = s_w[2];
    const uint32_t wbase = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
    const uint32_t slot = offsets[bid] + wbase + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
    if (slot >= cap) return;
    ivx_contact c;
    unsigned long long id = p.id_ab;  // contact_id_from_collidable_ids_and_indices(a, b, [0, i, j, k])
    id = splitmix(id ^ splitmix(0ull));
    id = splitmix(id ^ splitmix((unsigned long long)h.ijk[0]));
    id = splitmix(id ^ splitmix((unsigned long long)h.ijk[1]));
    id = splitmix(id ^ splitmix((unsigned long long)h.ijk[2]));
    c.id = id;
    c.body_a = p.body_a;
    c.body_b = p.body_b;
    c.position[0] = h.pos.x, c.position[1] = h.pos.y, c.position[2] = h.pos.z;
    c.normal[0] = h.nrm.x, c.normal[1] = h.nrm.y, c.normal[2] = h.nrm.z;
    c.depth = h.depth;
    c.restitution = p.restitution;
    c.static_friction = p.static_friction;
    c.dynamic_friction = p.dynamic_friction;
    c.flags = slot == 0 ? (uint32_t)IVX_CONTACT_MANIFOLD_START : 0u;  // one collision = one manifold
    c.reserved = 0;
    out[slot] = c;
}
__global__ __launch_bounds__(256) void k_mut_emit(MutEmitArgs a) { mut_emit_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_mut_emit_many, MutEmitArgs, mut_emit_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_mut_emit, k_mut_emit_many, MutEmitArgs, 256)
static_assert(sizeof(ScanCountsArgs) % 8 == 0 && sizeof(MutCountArgs) % 8 == 0 && sizeof(MutEmitArgs) % 8 == 0, "argument blocks travel as 8-byte words");

}  // namespace

static const int s_collide_many_registered = (ivx_many_register(IVX_MK_SCAN_COUNTS, many_scan_counts, sizeof(ScanCountsArgs)),
                                              ivx_many_register(IVX_MK_MUT_COUNT, many_mut_count, sizeof(MutCountArgs)),
                                              ivx_many_register(IVX_MK_MUT_EMIT, many_mut_emit, sizeof(MutEmitArgs)),
                                              ivx_many_register(IVX_MK_PROBE_SELECT_SMALL, many_probe_select_small, sizeof(ProbeSelectArgs)),
                                              ivx_many_register(IVX_MK_PROBE_SELECT_FULL, many_probe_select_full, sizeof(ProbeSelectArgs)),
                                              ivx_many_register(IVX_MK_PROBE_GATHER, many_probe_gather, sizeof(ProbeGatherArgs)), 0);

// d_slots = nullptr: every submesh (recompute); else the listed submesh slots (incremental sync), n_sub = their number. d_counts_host: optional
// host-mapped copy of the counts (ivx_collision_probes_sync_many). Inside a recorded batch the launches are captured.
int ivx_launch_probe_select(ivx_grid* g, uint32_t n_sub, uint32_t log2_bs, uint32_t* d_corner_list, uint32_t* d_sel, uint32_t* d_counts, uint32_t* d_offsets,
                            uint32_t* d_err, const uint32_t* d_slots, uint32_t* d_counts_host) {
    const uint32_t n_blocks = 1u << (3u * (4u - log2_bs));
    ProbeSelectArgs a;
    memset(&a, 0, sizeof(a));
    a.submeshes = g->submeshes, a.pos = g->positions, a.nrm = g->normals, a.idx = g->indices, a.corner_list = d_corner_list, a.sel = d_sel, a.counts = d_counts;
    a.counts_host = d_counts_host, a.err = d_err, a.slots = d_slots, a.log2_bs = log2_bs, a.inv_extent = 1.0f / g->extent;
    if (n_blocks <= PROBE_SMALLV && !ivx_many_try(g->ctx, g, IVX_MK_PROBE_SELECT_SMALL, n_sub, a))
        IVX_KLAUNCH((k_probe_select<PROBE_SMALLV, true>), dim3(n_sub), dim3(256), 0, g->ctx->stream, a);
    if (!ivx_many_try(g->ctx, g, IVX_MK_PROBE_SELECT_FULL, n_sub, a)) IVX_KLAUNCH((k_probe_select<PROBE_MAXV, false>), dim3(n_sub), dim3(256), 0, g->ctx->stream, a);
    if (d_offsets) {
        ScanCountsArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.counts = d_counts, sa.offsets = d_offsets, sa.n = n_sub;
        IVX_KLAUNCH(k_scan_counts, dim3(1), dim3(256), 0, g->ctx->stream, sa);
    }
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// copies the selected vertices to probe_points[d_offsets[record] ...]; d_entries (optional) gets one (chunk, first, end) entry per record
int ivx_launch_probe_gather(ivx_grid* g, uint32_t n_sub, uint32_t log2_bs, const uint32_t* d_sel, const uint32_t* d_counts, const uint32_t* d_offsets,
                            uint32_t* d_entries, const uint32_t* d_slots) {
    ProbeGatherArgs a;
    memset(&a, 0, sizeof(a));
    a.submeshes = g->submeshes, a.pos = g->positions, a.sel = d_sel, a.counts = d_counts, a.offsets = d_offsets, a.points = g->probe_points;
    a.probe_chunk = g->probe_chunk, a.entries = d_entries, a.slots = d_slots, a.n_blocks = 1u << (3u * (4u - log2_bs));
    if (!ivx_many_try(g->ctx, g, IVX_MK_PROBE_GATHER, n_sub, a)) IVX_KLAUNCH(k_probe_gather, dim3(n_sub), dim3(64), 0, g->ctx->stream, a);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// one pass of for_each_mutual_voxel_object_contact: the probes of `prober` against the SDF of `sampled`
int ivx_launch_mutual_pass(ivx_grid* prober, ivx_grid* sampled, const ivx_mutual_pass* h, uint32_t* d_counts, const uint32_t* d_offsets, ivx_contact* d_out,
                           uint32_t cap, int emit) {
    MutParams p;
    p.sdf = sampled->sdf;
    p.info = sampled->info;
    p.cy = sampled->cc[1];
    p.cz = sampled->cc[2];
    for (int d = 0; d < 3; ++d) {
        p.dims[d] = sampled->cc[d] * 16u;
        p.center[d] = h->center_s[d];
        p.t_s[d] = h->t_s[d];
        p.t_p[d] = h->t_p[d];
        p.box_lo[d] = h->box_lo[d];
        p.box_hi[d] = h->box_hi[d];
        p.clo[d] = h->clo[d];
        p.chi[d] = h->chi[d];
    }
    p.inv_s = 1.0f / sampled->extent;
    p.ext_s = sampled->extent;
    p.inv_p = 1.0f / prober->extent;
    p.q_s = Q4{h->q_s[0], h->q_s[1], h->q_s[2], h->q_s[3]};
    p.q_s_inv = Q4{-h->q_s[0], -h->q_s[1], -h->q_s[2], h->q_s[3]};
    p.q_p_inv = Q4{-h->q_p[0], -h->q_p[1], -h->q_p[2], h->q_p[3]};
    p.n_probes = prober->n_probe_points;
    p.negate = h->negate ? 1u : 0u;
    p.id_ab = h->id_ab;
    p.body_a = h->body_a;
    p.body_b = h->body_b;
    p.restitution = h->response[0];
    p.static_friction = h->response[1];
    p.dynamic_friction = h->response[2];
    const uint32_t n_wg = (prober->n_probe_points + 255u) / 256u;
    if (n_wg == 0) return IVX_OK;
    if (!emit) {
        MutCountArgs a;
        memset(&a, 0, sizeof(a));
        a.p = p, a.points = prober->probe_points, a.probe_chunk = prober->probe_chunk, a.counts = d_counts;
        if (!ivx_many_try(prober->ctx, prober, IVX_MK_MUT_COUNT, n_wg, a)) IVX_KLAUNCH(k_mut_count, dim3(n_wg), dim3(256), 0, prober->ctx->stream, a);
    } else {
        MutEmitArgs a;
        memset(&a, 0, sizeof(a));
        a.p = p, a.points = prober->probe_points, a.probe_chunk = prober->probe_chunk, a.offsets = d_offsets, a.out = d_out, a.cap = cap;
        if (!ivx_many_try(prober->ctx, prober, IVX_MK_MUT_EMIT, n_wg, a)) IVX_KLAUNCH(k_mut_emit, dim3(n_wg), dim3(256), 0, prober->ctx->stream, a);
    }
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_scan_counts(ivx_ctx* ctx, uint32_t n, const uint32_t* d_counts, uint32_t* d_offsets, uint32_t* d_total_out, const void* owner) {
    ScanCountsArgs a;
    memset(&a, 0, sizeof(a));
    a.counts = d_counts, a.offsets = d_offsets, a.total_out = d_total_out, a.n = n;
    if (owner && ivx_many_try(ctx, owner, IVX_MK_SCAN_COUNTS, 1u, a)) return IVX_OK;
    IVX_KLAUNCH(k_scan_counts, dim3(1), dim3(256), 0, ctx->stream, a);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
