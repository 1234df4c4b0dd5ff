// Voxel edit ops: an absorbing sphere or capsule eats into a voxel object — the per-frame mutators of the reference's deformable
// objects.
//
// Reference (engine/crates/impact_voxel/src):
//   apply_sphere_absorption                                   interaction/absorption.rs:801-844
//   VoxelAbsorbingSphere::compute_new_signed_distance         interaction/absorption.rs:170-180  (hard_sdf_subtraction, generation/sdf.rs:79-81)
//   Voxel::set_signed_distance                                lib.rs:451-461
//   modify_voxels_within_sphere + handle_chunk_voxels_modified  object/intersection.rs:283-395, 532-598
//   VoxelObjectInertialPropertyUpdater::remove_voxel          object/inertia.rs:377-394
//   apply_capsule_absorption                                  interaction/absorption.rs:846-889
//   modify_voxels_within_capsule                              object/intersection.rs:417-530
//   Capsule::trim_segment_outside_aab / compute_aabb / CapsulePointContainmentTester   impact_geometry/src/capsule.rs:132-250
//   AxisAlignedBox::find_contained_subsegment                 impact_geometry/src/axis_aligned_box.rs:385-415
// Per chunk of the touched chunk box: a Void chunk is skipped, a Uniform chunk becomes NonUniform (its 4096 voxels are written
// out) whether or not the sphere reaches a voxel of it, every voxel whose centre lies inside the influence sphere gets
// sd = quantise(max(sd, -(|p - c| - R))); a voxel that stops being negative is empty from then on and its mass moments and type
// are reported; a chunk left with only void voxels becomes Void. Derived state is recomputed afterwards by the ordinary sweep
// (derive.hip): it is a pure function of the voxels and the chunk kinds set here.
// The capsule differs in three places: the voxel ranges of a chunk come from the box of the capsule whose segment was clipped
// against the chunk box grown by the radius (a chunk the clipped capsule misses is left alone, Uniform or not), the distance is
// the one to the whole segment, and a voxel exactly on the boundary counts as inside (<=).
//
// One workgroup per chunk of the box, a thread owns a 16-voxel k-row.
#include <algorithm>

#include "chunk_passes.hpp"
#include "many.hpp"

namespace {

struct AbsorbParams {
    GridView g;
    uint32_t lo[3], cc[3];  // chunk box
    int32_t vlo[3], vhi[3];  // touched voxel ranges
    float c[3];  // sphere centre / capsule segment start
    float r2, r_sphere;
    int32_t capsule;  // 0 sphere, 1 capsule, 2 mutual absorption against another object's grid, 3 against a dense snapshot
    float seg[3], seg_over_len2[3], r_infl;
    // mutual absorption (interaction/absorption.rs:891-1094): where the other object's SDF is sampled
    const int8_t* o_sdf;          // mode 2: the other grid's sdf plane; mode 3: dense i8 snapshot over [s_lo, s_hi)
    const ivx_chunk_info* o_info;
    uint32_t o_cy, o_cz, o_dims[3];
    int32_t s_lo[3], s_hi[3];
    float q[4], t[3];             // transform_from_b_to_a; mode 2 applies its inverse, mode 3 applies it
    float ext_p, inv_s, ratio;    // own voxel extent, 1 / the other's, other's extent / own
    float smooth, quarter_inv;
};

__device__ __forceinline__ void rot3(const float q[4], const float v[3], float o[3]) {  // glam Quat::mul_vec3a
    const float b2 = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2], s1 = q[3] * q[3] - b2, s2 = ((v[0] * q[0] + v[1] * q[1]) + v[2] * q[2]) * 2.0f, s3 = q[3] * 2.0f;
    const float cx = q[1] * v[2] - v[1] * q[2], cy = q[2] * v[0] - v[2] * q[0], cz = q[0] * v[1] - v[0] * q[1];
    o[0] = (v[0] * s1 + q[0] * s2) + cx * s3;
    o[1] = (v[1] * s1 + q[1] * s2) + cy * s3;
    o[2] = (v[2] * s1 + q[2] * s2) + cz * s3;
}

// the other object's SDF at the centre of voxel (gi, gj, gk), in units of this object's voxels; false = the voxel is left alone
__device__ __forceinline__ bool mutual_other_sd(const AbsorbParams& p, int gi, int gj, int gk, float* inside_other) {
    const float c[3] = {((float)gi + 0.5f) * p.ext_p, ((float)gj + 0.5f) * p.ext_p, ((float)gk + 0.5f) * p.ext_p};
    float ps[3];
    if (p.capsule == 2) {  // inverse_voxel_extent_b * transform_from_b_to_a.inverse_transform_point(center_in_a)
        const float qi[4] = {-p.q[0], -p.q[1], -p.q[2], p.q[3]}, d[3] = {c[0] - p.t[0], c[1] - p.t[1], c[2] - p.t[2]};
        float r[3];
        rot3(qi, d, r);
        for (int a = 0; a < 3; ++a) ps[a] = p.inv_s * r[a];
    } else {  // inverse_voxel_extent_a * transform_from_b_to_a.transform_point(center_in_b)
        float r[3];
        rot3(p.q, c, r);
        for (int a = 0; a < 3; ++a) ps[a] = p.inv_s * (r[a] + p.t[a]);
    }
    float fl[3], off[3];
    for (int a = 0; a < 3; ++a) {
        const float lc = ps[a] - 0.5f;
        fl[a] = floorf(lc);
        off[a] = lc - fl[a];
    }
    float d[8];
    if (p.capsule == 2) {  // sample_voxel_object_sdf (object/sdf.rs:636-675)
        bool outside = ((__float_as_uint(fl[0]) | __float_as_uint(fl[1]) | __float_as_uint(fl[2])) & 0x80000000u) != 0;
        uint32_t l[3];
        for (int a = 0; a < 3; ++a) {
            const uint32_t u = (uint32_t)fl[a];  // saturating
            l[a] = u > 0x7FFFFFF0u ? 0x7FFFFFF0u : u;
            outside |= l[a] + 1u >= p.o_dims[a];
        }
        if (outside) {
            *inside_other = (127.0f * 0.02f) * p.ratio;
            return true;
        }
#pragma unroll
        for (int cn = 0; cn < 8; ++cn) {
            const uint32_t i = l[0] + ((cn >> 2) & 1), j = l[1] + ((cn >> 1) & 1), k = l[2] + (cn & 1);
            const uint32_t chunk = ((i >> 4) * p.o_cy + (j >> 4)) * p.o_cz + (k >> 4);
            const uint32_t kind = p.o_info[chunk].kind;
            const int sd = kind == KIND_NONUNIFORM ? (int)p.o_sdf[(size_t)chunk * IVX_CHUNK_VOXELS + (((i & 15u) << 8) | ((j & 15u) << 4) | (k & 15u))]
                                                   : (kind == KIND_UNIFORM ? -128 : 127);
            d[cn] = (float)sd * 0.02f;
        }
    } else {  // the snapshot of A's distances as they were before A was modified
        long long l[3];
        for (int a = 0; a < 3; ++a) {
            l[a] = fl[a] >= 9.0e18f ? 0x7FFFFFFFFFFFFFF0ll : (fl[a] <= -9.0e18f ? -0x7FFFFFFFFFFFFFF0ll : (long long)fl[a]);  // `as isize`
            if (l[a] < (long long)p.s_lo[a] || l[a] + 1 >= (long long)p.s_hi[a]) return false;
        }
        const size_t ny = (size_t)(p.s_hi[1] - p.s_lo[1]), nz = (size_t)(p.s_hi[2] - p.s_lo[2]);
#pragma unroll
        for (int cn = 0; cn < 8; ++cn) {
            const size_t i = (size_t)(l[0] - p.s_lo[0]) + ((cn >> 2) & 1), j = (size_t)(l[1] - p.s_lo[1]) + ((cn >> 1) & 1), k = (size_t)(l[2] - p.s_lo[2]) + (cn & 1);
            d[cn] = (float)(int)p.o_sdf[(i * ny + j) * nz + k] * 0.02f;
        }
    }
    const float rx = 1.0f - off[0], ry = 1.0f - off[1], rz = 1.0f - off[2];  // evaluate_sdf_from_corner_samples (object/sdf.rs:579-597)
    const float d00 = d[0] * rx + d[4] * off[0], d01 = d[1] * rx + d[5] * off[0], d10 = d[2] * rx + d[6] * off[0], d11 = d[3] * rx + d[7] * off[0];
    const float d0 = d00 * ry + d10 * off[1], d1 = d01 * ry + d11 * off[1];
    *inside_other = (d0 * rz + d1 * off[2]) * p.ratio;
    return true;
}

// compute_subtracted_signed_distance (interaction/absorption.rs:1081-1094) with sdf_subtraction (generation/sdf.rs:52-102)
__device__ __forceinline__ float subtracted_sd(const AbsorbParams& p, float sd, float inside_other) {
    const float inter = inside_other > sd ? inside_other : sd;
    if (p.smooth == 0.0f) return (-inter > sd) ? -inter : sd;
    const float a = -sd, diff = fabsf(a - inter), hh = p.smooth - diff, h = hh > 0.0f ? hh : 0.0f;
    const float mn = inter < a ? inter : a;
    return -(mn - (h * h) * p.quarter_inv);
}

__device__ __forceinline__ int quantise(float v) {  // VoxelSignedDistance::from_f32 (lib.rs:197-201)
    float s = v * 50.0f;
    if (s != s) return 0;
    s = s < -128.0f ? -128.0f : (s > 127.0f ? 127.0f : s);
    return (int)s;
}

struct AbsorbArgs {
    AbsorbParams p;
    int8_t* sdf;
    uint8_t* type;
    ivx_chunk_info* info;
    const float* dens;
    double* removed10;
    uint32_t* by_type;
    uint32_t* counters;
    uint32_t* touched_ranges;
    uint32_t* zero16;
};
__device__ __forceinline__ void absorb_body(const AbsorbArgs& a_, uint32_t b, uint32_t) {
    const AbsorbParams& p = a_.p;
    int8_t* __restrict__ sdf = a_.sdf;
    uint8_t* __restrict__ type = a_.type;
    ivx_chunk_info* __restrict__ info = a_.info;
    const float* __restrict__ dens = a_.dens;
    double* __restrict__ removed10 = a_.removed10;
    uint32_t* __restrict__ by_type = a_.by_type;
    uint32_t* __restrict__ counters = a_.counters;
    uint32_t* __restrict__ touched_ranges = a_.touched_ranges;
    uint32_t* __restrict__ zero16 = a_.zero16;
    __shared__ float s_dens[256];
    __shared__ double s_red[16][10];
    const uint32_t tid = threadIdx.x;
    // (the region scalars the sweep behind this launch starts from: sixteen words the edit path would otherwise spend a stream operation on)
    if (zero16 && b == 0u && tid < 16u) zero16[tid] = 0u;
    const uint32_t bk = b % p.cc[2], bj = (b / p.cc[2]) % p.cc[1], bi = b / (p.cc[2] * p.cc[1]);
    const uint32_t ci = p.lo[0] + bi, cj = p.lo[1] + bj, ck = p.lo[2] + bk;
    const uint32_t chunk = (ci * p.g.cy + cj) * p.g.cz + ck;
    // the voxel ranges of this chunk the shape may reach (the same in every thread)
    const int cbase[3] = {(int)(ci * 16u), (int)(cj * 16u), (int)(ck * 16u)};
    int rlo[3], rhi[3];
    if (p.capsule == 1) {
        float t_min = 0.0f, t_max = 1.0f;
        bool none = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float blo = (float)cbase[d] - p.r_infl, bhi = (float)(cbase[d] + 16) + p.r_infl;
            if (fabsf(p.seg[d]) > 1e-8f) {
                const float recip = 1.0f / p.seg[d];
                const float t1 = (blo - p.c[d]) * recip, t2 = (bhi - p.c[d]) * recip;
                const float te = t1 < t2 ? t1 : t2, tx = t1 < t2 ? t2 : t1;
                t_min = te > t_min ? te : t_min;  // f32::max / f32::min (no NaN: recip is finite)
                t_max = tx < t_max ? tx : t_max;
            } else if (p.c[d] < blo || p.c[d] > bhi) {
                none = true;
            }
        }
        if (none || !(t_min <= t_max)) return;
        bool empty = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float ts = p.c[d] + p.seg[d] * t_min;
            const float tv = p.seg[d] * (t_max - t_min);
            const float te = ts + tv;
            const float a0 = ts - p.r_infl, a1 = te - p.r_infl, b0 = ts + p.r_infl, b1 = te + p.r_infl;
            const float lo_f = a1 < a0 ? a1 : a0, hi_f = b1 > b0 ? b1 : b0;
            const float fl = floorf(lo_f), ce = ceilf(hi_f);
            // (`as usize` saturates; beyond the chunk either way the clamp below decides)
            const int s_i = fl > 0.0f ? (fl < 2.0e9f ? (int)fl : 2000000000) : 0, e_i = ce > 0.0f ? (ce < 2.0e9f ? (int)ce : 2000000000) : 0;
            rlo[d] = s_i > cbase[d] ? s_i : cbase[d];
            rhi[d] = e_i < cbase[d] + 16 ? e_i : cbase[d] + 16;
            empty |= rlo[d] >= rhi[d];
        }
        if (empty) return;
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            rlo[d] = p.vlo[d] > cbase[d] ? p.vlo[d] : cbase[d];
            rhi[d] = p.vhi[d] < cbase[d] + 16 ? p.vhi[d] : cbase[d] + 16;
        }
    }
    const ivx_chunk_info rec = info[chunk];
    if (rec.kind == KIND_VOID) return;
    s_dens[tid] = dens[tid];
    const bool was_uniform = rec.kind != KIND_NONUNIFORM;
    const size_t o = (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)tid * 16;
    uint32_t sw[4], tw[4];
    if (was_uniform) {  // convert_to_non_uniform_if_uniform (object.rs:2530-2550)
        sw[0] = sw[1] = sw[2] = sw[3] = 0x80808080u;
        tw[0] = tw[1] = tw[2] = tw[3] = (uint32_t)rec.uniform_type * 0x01010101u;
    } else {
        const uint4 s4 = *reinterpret_cast<const uint4*>(sdf + o), t4 = *reinterpret_cast<const uint4*>(type + o);
        sw[0] = s4.x, sw[1] = s4.y, sw[2] = s4.z, sw[3] = s4.w;
        tw[0] = t4.x, tw[1] = t4.y, tw[2] = t4.z, tw[3] = t4.w;
    }
    const int gi = (int)(ci * 16u + (tid >> 4)), gj = (int)(cj * 16u + (tid & 15u));
    const bool row_in = gi >= rlo[0] && gi < rhi[0] && gj >= rlo[1] && gj < rhi[1];
    const float px = (float)gi + 0.5f, py = (float)gj + 0.5f;
    const float dx = px - p.c[0], dy = py - p.c[1];
    uint32_t emptied = 0;  // bit k: the voxel was non-empty and is empty now
    bool any_inside = false, changed = false;
    uint32_t non_empty = 0, non_void = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int gk = (int)(ck * 16u) + k;
        int sd = (int)(int8_t)((sw[k >> 2] >> (8 * (k & 3))) & 0xFFu);
        if (row_in && gk >= rlo[2] && gk < rhi[2]) {
            const float pz = (float)gk + 0.5f;
            const float dz = pz - p.c[2];
            float d2 = 0.0f, other = 0.0f;
            bool inside;
            if (p.capsule >= 2) {  // mutual absorption: every voxel of the ranges that is not maximally outside
                inside = sd != 127 && mutual_other_sd(p, gi, gj, gk, &other);
            } else if (p.capsule) {  // CapsulePointContainmentTester::shortest_squared_distance_from_point_to_segment
                float t = (dx * p.seg_over_len2[0] + dy * p.seg_over_len2[1]) + dz * p.seg_over_len2[2];
                t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);  // f32::clamp
                const float ex = px - (p.c[0] + p.seg[0] * t), ey = py - (p.c[1] + p.seg[1] * t), ez = pz - (p.c[2] + p.seg[2] * t);
                d2 = (ex * ex + ey * ey) + ez * ez;
                inside = d2 <= p.r2;
            } else {
                d2 = (dx * dx + dy * dy) + dz * dz;
                inside = d2 < p.r2;
            }
            if (inside) {
                any_inside = true;
                const float old = (float)sd * 0.02f;
                float nv;
                if (p.capsule >= 2) {
                    nv = subtracted_sd(p, old, other);
                } else {
                    const float neg = -(sqrtf(d2) - p.r_sphere);
                    nv = (neg > old) ? neg : old;  // f32::max (no NaN here)
                }
                const int q = quantise(nv);
                if (sd < 0 && q >= 0) emptied |= 1u << k;
                if (q != sd) {
                    changed = true;
                    sd = q;
                    sw[k >> 2] = (sw[k >> 2] & ~(0xFFu << (8 * (k & 3)))) | ((uint32_t)(q & 0xFF) << (8 * (k & 3)));
                }
            }
        }
        non_empty |= sd < 0 ? 1u : 0u;
        non_void |= sd <= SD_VOID_LIMIT ? 1u : 0u;
    }
    const int touched = __syncthreads_or(any_inside ? 1 : 0);
    const int has_non_empty = __syncthreads_or((int)non_empty);
    const int has_non_void = __syncthreads_or((int)non_void);
    const int any_emptied = __syncthreads_or(emptied != 0);
    const bool becomes_void = touched && !has_non_empty && !has_non_void;
    if (!becomes_void && (was_uniform || changed)) {
        *reinterpret_cast<uint4*>(sdf + o) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
        if (was_uniform) *reinterpret_cast<uint4*>(type + o) = make_uint4(tw[0], tw[1], tw[2], tw[3]);
    }
    if (any_emptied) {
        // what the inertial property updater removes (inertia.rs:377-394), in the integer form of inertia.hip
        double m[10];
        chunk_moments_rows(tid, emptied, tw, s_dens, s_red, gi, gj, (int)(ck * 16u), m);  // (every thread passes its own m; tid < 10 hold the sums)
        if (tid < 10) atomicAdd(&removed10[tid], m[tid]);
        // the tracker's per-type counts: one global add per (chunk, type) — a chunk holds a few types at most, and an add per
        // voxel on the one counter of a single-material body took 2 ms for a 100 k-voxel bite. Each round the workgroup agrees on
        // the type of the first voxel still uncounted and counts all emptied voxels of that type.
        __shared__ uint32_t s_pick, s_cnt;
        uint32_t rem = emptied;
        for (int guard = 0; guard < 256; ++guard) {
            if (tid == 0) {
                s_pick = 0xFFFFFFFFu;
                s_cnt = 0;
            }
            __syncthreads();
            if (rem) {
                const int k = __ffs(rem) - 1;
                atomicMin(&s_pick, (tid << 8) | ((tw[k >> 2] >> (8 * (k & 3))) & 0xFFu));
            }
            __syncthreads();
            const uint32_t pick = s_pick;
            if (pick == 0xFFFFFFFFu) break;  // (the same value in every thread)
            const uint32_t t = pick & 0xFFu;
            uint32_t mine = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (((rem >> k) & 1u) && ((tw[k >> 2] >> (8 * (k & 3))) & 0xFFu) == t) {
                    mine += 1;
                    rem &= ~(1u << k);
                }
            const uint32_t wsum = ivx_wave_sum(mine);
            if ((tid & 63u) == 0 && wsum) atomicAdd(&s_cnt, wsum);
            __syncthreads();
            if (tid == 0) atomicAdd(&by_type[t], s_cnt);
            __syncthreads();
        }
    }
    if (tid == 0) {
        ivx_chunk_info out = rec;
        if (becomes_void) {
            out.kind = out.gen_kind = KIND_VOID;
            out.flags = 0;
            out.uniform_type = 0;
            atomicAdd(&counters[1], 1u);
        } else {
            out.kind = out.gen_kind = KIND_NONUNIFORM;  // touched or not, a chunk of the box is NonUniform from now on
            out.flags = has_non_empty ? 0 : (uint8_t)CF_ONLY_EMPTY;
            out.uniform_type = 0;
        }
        info[chunk] = out;
        if (touched) {  // with the voxel ranges handle_chunk_voxels_modified sees, chunk-relative: lo 0..15, hi-1 0..15
            touched_ranges[b] = 0x80000000u | (uint32_t)(rlo[0] - cbase[0]) | ((uint32_t)(rlo[1] - cbase[1]) << 4) | ((uint32_t)(rlo[2] - cbase[2]) << 8) |
                                    ((uint32_t)(rhi[0] - 1 - cbase[0]) << 12) | ((uint32_t)(rhi[1] - 1 - cbase[1]) << 16) |
                                    ((uint32_t)(rhi[2] - 1 - cbase[2]) << 20);
            atomicAdd(&counters[0], 1u);
        }
    }
}

__global__ __launch_bounds__(256) void k_absorb(AbsorbArgs a) { absorb_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_absorb_many, AbsorbArgs, absorb_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_absorb, k_absorb_many, AbsorbArgs, 256)
static_assert(sizeof(AbsorbArgs) % 8 == 0, "argument blocks travel as 8-byte words");
static const int s_many_registered_absorb = (ivx_many_register(IVX_MK_ABSORB, many_absorb, sizeof(AbsorbArgs)), 0);

// dense copy of the object's distances over a voxel box (Void chunks read 127, Uniform ones -128), x-major
__global__ __launch_bounds__(256) void k_sdf_snapshot(GridView g, const int8_t* __restrict__ sdf, int3 lo, int3 n, int8_t* __restrict__ out) {
    const size_t total = (size_t)n.x * n.y * n.z;
    for (size_t e = (size_t)blockIdx.x * 256u + threadIdx.x; e < total; e += (size_t)gridDim.x * 256u) {
        const uint32_t k = (uint32_t)(e % n.z) + lo.z, j = (uint32_t)((e / n.z) % n.y) + lo.y, i = (uint32_t)(e / ((size_t)n.z * n.y)) + lo.x;
        const uint32_t chunk = ((i >> 4) * g.cy + (j >> 4)) * g.cz + (k >> 4);
        const uint32_t kind = g.info[chunk].kind;
        out[e] = kind == KIND_NONUNIFORM ? sdf[(size_t)chunk * IVX_CHUNK_VOXELS + (((i & 15u) << 8) | ((j & 15u) << 4) | (k & 15u))]
                                         : (int8_t)(kind == KIND_UNIFORM ? -128 : 127);
    }
}

}  // namespace

int ivx_launch_sdf_snapshot(ivx_grid* g, const int32_t lo[3], const int32_t hi[3], int8_t* d_out) {
    const int3 l = make_int3(lo[0], lo[1], lo[2]), n = make_int3(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
    const size_t total = (size_t)n.x * n.y * n.z;
    if (total == 0) return IVX_OK;
    const uint32_t wgs = (uint32_t)std::min<size_t>((total + 255) / 256, 65535u * 16u);
    IVX_KLAUNCH(k_sdf_snapshot, dim3(wgs), dim3(256), 0, g->ctx->stream, ivx_view(g), g->sdf, l, n, d_out);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

// one side of apply_mutual_absorption: the voxels of `g` inside [vlo, vhi) against the other object's SDF — its live grid (from_snapshot 0, the
// A side) or the dense snapshot of A taken before A was modified (from_snapshot 1, the B side)
int ivx_launch_absorb_mutual(ivx_grid* g, int from_snapshot, const uint32_t lo[3], const uint32_t cc[3], const int32_t vlo[3], const int32_t vhi[3],
                             ivx_grid* other, const int8_t* d_snapshot, const int32_t s_lo[3], const int32_t s_hi[3], const float q_ba[4],
                             const float t_ba[3], float smoothness, const float* d_dens, double* d_removed10, uint32_t* d_by_type, uint32_t* d_counters,
                             uint32_t* d_touched) {
    uint32_t* const d_zero16 = nullptr;
    ivx_planes_touched(g);
    AbsorbParams p;
    p.g = ivx_view(g);
    p.capsule = from_snapshot ? 3 : 2;
    for (int d = 0; d < 3; ++d) {
        p.lo[d] = lo[d];
        p.cc[d] = cc[d];
        p.vlo[d] = vlo[d];
        p.vhi[d] = vhi[d];
        p.c[d] = p.seg[d] = p.seg_over_len2[d] = 0.0f;
        p.o_dims[d] = other->cc[d] * 16u;
        p.s_lo[d] = s_lo[d];
        p.s_hi[d] = s_hi[d];
        p.t[d] = t_ba[d];
    }
    for (int d = 0; d < 4; ++d) p.q[d] = q_ba[d];
    p.r2 = p.r_infl = p.r_sphere = 0.0f;
    p.o_sdf = from_snapshot ? d_snapshot : other->sdf;
    p.o_info = other->info;
    p.o_cy = other->cc[1];
    p.o_cz = other->cc[2];
    p.ext_p = g->extent;
    p.inv_s = 1.0f / other->extent;
    p.ratio = other->extent * (1.0f / g->extent);
    p.smooth = smoothness;
    p.quarter_inv = 0.25f / smoothness;
    {
        AbsorbArgs aa;
        memset(&aa, 0, sizeof(aa));
        aa.p = p, aa.sdf = g->sdf, aa.type = g->type, aa.info = g->info, aa.dens = d_dens, aa.removed10 = d_removed10, aa.by_type = d_by_type;
        aa.counters = d_counters, aa.touched_ranges = d_touched, aa.zero16 = d_zero16;
        const uint32_t blocks = cc[0] * cc[1] * cc[2];
        if (!ivx_many_try(g->ctx, g, IVX_MK_ABSORB, blocks, aa)) IVX_KLAUNCH(k_absorb, dim3(blocks), dim3(256), 0, g->ctx->stream, aa);
    }
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_absorb(ivx_grid* g, int capsule, const uint32_t lo[3], const uint32_t cc[3], const int32_t vlo[3], const int32_t vhi[3], const float c[3],
                      const float seg[3], float influence_radius, float shape_radius, const float* d_dens, double* d_removed10, uint32_t* d_by_type,
                      uint32_t* d_counters, uint32_t* d_touched, uint32_t* d_zero16) {
    ivx_planes_touched(g);
    AbsorbParams p;
    p.g = ivx_view(g);
    p.capsule = capsule;
    const float len2 = capsule ? (seg[0] * seg[0] + seg[1] * seg[1]) + seg[2] * seg[2] : 0.0f;
    const float inv = len2 > 1e-8f ? 1.0f / len2 : 0.0f;  // impact_math Div<f32>: multiply by the reciprocal
    for (int d = 0; d < 3; ++d) {
        p.lo[d] = lo[d];
        p.cc[d] = cc[d];
        p.vlo[d] = vlo[d];
        p.vhi[d] = vhi[d];
        p.c[d] = c[d];
        p.seg[d] = capsule ? seg[d] : 0.0f;
        p.seg_over_len2[d] = len2 > 1e-8f ? p.seg[d] * inv : 0.0f;
    }
    p.r2 = influence_radius * influence_radius;
    p.r_infl = influence_radius;
    p.r_sphere = shape_radius;
    p.o_sdf = nullptr;
    p.o_info = nullptr;
    p.o_cy = p.o_cz = 0;
    for (int d = 0; d < 3; ++d) p.o_dims[d] = 0, p.s_lo[d] = p.s_hi[d] = 0, p.t[d] = 0.0f;
    p.q[0] = p.q[1] = p.q[2] = 0.0f, p.q[3] = 1.0f;
    p.ext_p = p.inv_s = p.ratio = 1.0f;
    p.smooth = p.quarter_inv = 0.0f;
    {
        AbsorbArgs aa;
        memset(&aa, 0, sizeof(aa));
        aa.p = p, aa.sdf = g->sdf, aa.type = g->type, aa.info = g->info, aa.dens = d_dens, aa.removed10 = d_removed10, aa.by_type = d_by_type;
        aa.counters = d_counters, aa.touched_ranges = d_touched, aa.zero16 = d_zero16;
        const uint32_t blocks = cc[0] * cc[1] * cc[2];
        if (!ivx_many_try(g->ctx, g, IVX_MK_ABSORB, blocks, aa)) IVX_KLAUNCH(k_absorb, dim3(blocks), dim3(256), 0, g->ctx->stream, aa);
    }
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
