// Voxel contact generation (first part): contacts between a sphere, plane or capsule collidable and the surface
// voxels of a voxel object — the step immediately before the constraint solver.
//
// Reference (engine/crates):
//   for_each_sphere_voxel_object_contact                    impact_voxel/src/collidable.rs:1098-1127
//   for_each_surface_voxel_maybe_intersecting_sphere        impact_voxel/src/object/intersection.rs:51-60, 97-151
//   voxel_ranges_touching_aab                               impact_voxel/src/object/intersection.rs:766-782
//   VoxelFlags::placement                                   impact_voxel/src/lib.rs:330-342
//   compute_voxel_radius                                    impact_voxel/src/collidable.rs:1453-1455
//   determine_sphere_sphere_contact_geometry                impact_physics/src/collision/collidable/sphere.rs:105-136
//   ContactID::from_two_u64_and_n_indices                   impact_physics/src/constraint/contact.rs:180-199
//   for_each_capsule_voxel_object_contact                   impact_voxel/src/collidable.rs:1257-1286
//   determine_capsule_sphere_contact_geometry               impact_physics/src/collision/collidable/capsule.rs:212-270
//   parameter_of_closest_point_on_line_segment_to_point     impact_geometry/src/line.rs:26-45
// Every non-empty voxel with fewer than six neighbours inside the touched voxel ranges is a small sphere (radius = -sd * extent)
// tested against the collidable. The reference emits contacts in traversal order (chunks i,j,k, then voxels i,j,k), and the
// solver's result depends on that order, so the emit pass is an ordered compaction: counts per chunk, a scan over the chunks
// of the box, then thread-ordered prefix sums inside each chunk.
#include "ivx_internal.hpp"

namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
struct Q4 {
    float x, y, z, w;
};
// glam Quat::mul_vec3a
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {
    const V3 b = mk(q.x, q.y, q.z);
    const float b2 = dot(b, b);
    return (v * (q.w * q.w - b2) + b * (dot(v, b) * 2.0f)) + cross(b, v) * (q.w * 2.0f);
}
// impact_math/src/random/splitmix.rs:4-10
__device__ __forceinline__ unsigned long long splitmix(unsigned long long state) {
    state += 0x9E3779B97F4A7C15ull;
    unsigned long long z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct SvcParams {
    GridView g;
    uint32_t lo[3], cc[3];   // chunk box
    int32_t vlo[3], vhi[3];  // touched voxel ranges
    Q4 q_inv;                // inverse rotation of transform_to_object_space
    float t[3];              // its translation
    uint32_t mode;           // 0: sphere collidable (c, r); 1: plane collidable (unit normal c, displacement r), Corner voxels only;
                             // 2: capsule collidable (segment start c, segment vector v, radius r)
    float c[3], r;           // the collidable (world space)
    float v[3], len2;
    float extent;
    unsigned long long id_ab;  // splitmix(a ^ splitmix(b))
    uint32_t body_a, body_b;
    float restitution, static_friction, dynamic_friction;
};

struct Hit {
    V3 pos, nrm;
    float depth;
};

// the thread's row (i, j): bit k of the result = voxel k yields a contact; hits[] filled for those (emit pass only)
template <bool EMIT>
__device__ __forceinline__ uint32_t row_contacts(const SvcParams& p, const int8_t* sdf, const uint8_t* flags, uint32_t chunk, uint32_t ci, uint32_t cj,
                                                 uint32_t ck, uint32_t tid, Hit* hits) {
    const int gi = (int)(ci * 16u + (tid >> 4)), gj = (int)(cj * 16u + (tid & 15u));
    if (gi < p.vlo[0] || gi >= p.vhi[0] || gj < p.vlo[1] || gj >= p.vhi[1]) return 0u;
    const size_t o = (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)tid * 16;
    const uint4 s4 = *reinterpret_cast<const uint4*>(sdf + o), f4 = *reinterpret_cast<const uint4*>(flags + o);
    const uint32_t sw[4] = {s4.x, s4.y, s4.z, s4.w}, fw[4] = {f4.x, f4.y, f4.z, f4.w};
    uint32_t mask = 0;
    const V3 c = mk(p.c[0], p.c[1], p.c[2]);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int gk = (int)(ck * 16u) + k;
        if (gk < p.vlo[2] || gk >= p.vhi[2]) continue;
        const uint32_t f = (fw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        // empty, or Interior (six neighbours, lib.rs:330-342); a plane only needs the Corner voxels (at most three, collidable.rs:1187-1191)
        if ((f & VF_EMPTY) || __popc(f & 0xFCu) > (p.mode == 1u ? 3 : 5)) continue;
        const int sd = (int)(int8_t)((sw[k >> 2] >> (8 * (k & 3))) & 0xFFu);
        const V3 p_obj = mk(((float)gi + 0.5f) * p.extent, ((float)gj + 0.5f) * p.extent, ((float)gk + 0.5f) * p.extent);
        const V3 pw = qrot(p.q_inv, p_obj - mk(p.t[0], p.t[1], p.t[2]));  // inverse_transform_point
        const float vr = -((float)sd * 0.02f) * p.extent;                  // compute_voxel_radius
        if (p.mode == 2u) {  // determine_capsule_sphere_contact_geometry(capsule, voxel_sphere)
            const V3 sv = mk(p.v[0], p.v[1], p.v[2]);
            float param = 0.0f;
            if (!(p.len2 <= 1e-8f)) {
                param = dot(sv, pw - c) / p.len2;
                param = param < 0.0f ? 0.0f : (param > 1.0f ? 1.0f : param);  // f32::clamp
            }
            const V3 disp = pw - (c + sv * param);
            const float d2 = dot(disp, disp), maxd = vr + p.r;
            if (d2 > maxd * maxd) continue;
            mask |= 1u << k;
            if (EMIT) {
                const float dist = sqrtf(d2);
                V3 cn;
                float pen;
                if (dist > 1e-8f) {
                    cn = disp * (1.0f / dist);
                    pen = maxd - dist;
                } else {  // the voxel centre lies on the segment: any normal to the segment (glam Vec3A::any_orthogonal_vector)
                    const V3 o = fabsf(sv.x) > fabsf(sv.y) ? mk(-sv.z, 0.0f, sv.x) : mk(0.0f, sv.z, -sv.y);
                    const float n2 = dot(o, o);
                    if (n2 > 1e-8f * 1e-8f) {
                        const float nn = sqrtf(n2);
                        cn = mk(o.x / nn, o.y / nn, o.z / nn);
                    } else {
                        cn = mk(0.0f, 0.0f, 1.0f);
                    }
                    pen = maxd;
                }
                const V3 n = mk(-cn.x, -cn.y, -cn.z);
                hits[k].nrm = n;
                hits[k].pos = pw + n * vr;
                hits[k].depth = pen > 0.0f ? pen : 0.0f;
            }
            continue;
        }
        if (p.mode) {  // determine_sphere_plane_contact_geometry(voxel_sphere, plane) (sphere.rs:138-160)
            const float sdist = dot(c, pw) - p.r;
            const float pen = vr - sdist;
            if (pen < 0.0f) continue;
            mask |= 1u << k;
            if (EMIT) {
                hits[k].nrm = c;
                hits[k].pos = pw - c * sdist;
                hits[k].depth = pen;
            }
            continue;
        }
        // determine_sphere_sphere_contact_geometry(sphere, voxel_sphere)
        const V3 d = c - pw;
        const float d2 = dot(d, d), maxd = p.r + vr;
        if (d2 > maxd * maxd) continue;
        mask |= 1u << k;
        if (EMIT) {
            const float dist = sqrtf(d2);
            const float inv = 1.0f / dist;
            const V3 n = dist > 1e-8f ? d * inv : mk(0.0f, 0.0f, 1.0f);
            hits[k].nrm = n;
            hits[k].pos = pw + n * vr;
            const float pen = maxd - dist;
            hits[k].depth = pen > 0.0f ? pen : 0.0f;
        }
    }
    return mask;
}

__device__ __forceinline__ uint32_t box_chunk(const SvcParams& p, uint32_t b, uint32_t& ci, uint32_t& cj, uint32_t& ck) {
    const uint32_t bk = b % p.cc[2], bj = (b / p.cc[2]) % p.cc[1], bi = b / (p.cc[2] * p.cc[1]);
    ci = p.lo[0] + bi, cj = p.lo[1] + bj, ck = p.lo[2] + bk;
    return (ci * p.g.cy + cj) * p.g.cz + ck;
}

// (the three passes as bodies over an argument block: the single-object kernels and their many-object twins, many.hpp, run the same code)
struct SvcCountArgs {
    SvcParams p;
    const uint8_t* flags;
    uint32_t* counts;
};
struct SvcScanArgs {
    const uint32_t* counts;
    uint32_t* offsets;
    uint32_t* total;
    uint32_t n, pad;
};
struct SvcEmitArgs {
    SvcParams p;
    const uint8_t* flags;
    const uint32_t* offsets;
    ivx_contact* out;
    uint32_t cap, pad;
};
__device__ __forceinline__ void svc_count_body(const SvcCountArgs& a, uint32_t bid, uint32_t) {
    const SvcParams& p = a.p;
    const uint8_t* __restrict__ flags = a.flags;
    uint32_t* __restrict__ counts = a.counts;
    __shared__ uint32_t s_w[4];
    uint32_t ci, cj, ck;
    const uint32_t chunk = box_chunk(p, bid, ci, cj, ck);
    uint32_t n = 0;
    if (p.g.info[chunk].kind == KIND_NONUNIFORM)  // only non-uniform chunks can have surface voxels
        n = __popc(row_contacts<false>(p, p.g.sdf, flags, chunk, ci, cj, ck, threadIdx.x, nullptr));
    n = ivx_wave_sum(n);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[bid] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
__global__ __launch_bounds__(256) void k_svc_count(SvcCountArgs a) { svc_count_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_svc_count_many, SvcCountArgs, svc_count_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_svc_count, k_svc_count_many, SvcCountArgs, 256)

// exclusive scan over the chunks of the box (one workgroup; the box of a collidable is a few dozen chunks)
__device__ __forceinline__ void svc_scan_body(const SvcScanArgs& a, uint32_t, uint32_t) {
    const uint32_t n = a.n;
    const uint32_t* __restrict__ counts = a.counts;
    uint32_t* __restrict__ offsets = a.offsets;
    uint32_t* __restrict__ total = a.total;
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n; b0 += 256u) {
        const uint32_t b = b0 + tid;
        const uint32_t v = b < n ? counts[b] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= (uint32_t)o) incl += t;
        }
        if (lane == 63u) s_w[wave] = incl;
        __syncthreads();
        const uint32_t w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
        const uint32_t wbase = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
        if (b < n) offsets[b] = s_carry + wbase + incl - v;
        __syncthreads();
        if (tid == 0) s_carry += (w0 + w1) + (w2 + w3);
        __syncthreads();
    }
    if (tid == 0) *total = s_carry;
}
__global__ __launch_bounds__(256) void k_svc_scan(SvcScanArgs a) { svc_scan_body(a, 0u, 1u); }
IVX_MANY_TWIN(k_svc_scan_many, SvcScanArgs, svc_scan_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_svc_scan, k_svc_scan_many, SvcScanArgs, 256)

__device__ __forceinline__ void svc_emit_body(const SvcEmitArgs& a, uint32_t bid, uint32_t) {
    const SvcParams& p = a.p;
    const uint8_t* __restrict__ flags = a.flags;
    const uint32_t* __restrict__ offsets = a.offsets;
    const uint32_t cap = a.cap;
    ivx_contact* __restrict__ out = a.out;
    __shared__ uint32_t s_w[4];
    uint32_t ci, cj, ck;
    const uint32_t chunk = box_chunk(p, bid, ci, cj, ck);
    if (p.g.info[chunk].kind != KIND_NONUNIFORM) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    Hit hits[16];
    const uint32_t mask = row_contacts<true>(p, p.g.sdf, flags, chunk, ci, cj, ck, tid, hits);
    const uint32_t v = __popc(mask);
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o) incl += t;
    }
    if (lane == 63u) s_w[wave] = incl;
    __syncthreads();
    const uint32_t w0 = s_w[0], w1 = s_w[1], w2 = s_w[2];
    const uint32_t wbase = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
    uint32_t slot = offsets[bid] + wbase + incl - v;
    const unsigned long long gi = ci * 16u + (tid >> 4), gj = cj * 16u + (tid & 15u);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (!((mask >> k) & 1u)) continue;
        if (slot < cap) {
            ivx_contact c;
            unsigned long long id = p.id_ab;
            id = splitmix(id ^ splitmix(gi));
            id = splitmix(id ^ splitmix(gj));
            id = splitmix(id ^ splitmix((unsigned long long)(ck * 16u + (uint32_t)k)));
            c.id = id;
            c.body_a = p.body_a;
            c.body_b = p.body_b;
            c.position[0] = hits[k].pos.x, c.position[1] = hits[k].pos.y, c.position[2] = hits[k].pos.z;
            c.normal[0] = hits[k].nrm.x, c.normal[1] = hits[k].nrm.y, c.normal[2] = hits[k].nrm.z;
            c.depth = hits[k].depth;
#ifdef IVX_MUTATION_CHECK  // tools/mutation_check.sh (see collide.hip)
            c.depth = __uint_as_float(__float_as_uint(c.depth) + 1u);
#endif
            c.restitution = p.restitution;
            c.static_friction = p.static_friction;
            c.dynamic_friction = p.dynamic_friction;
            c.flags = slot == 0 ? (uint32_t)IVX_CONTACT_MANIFOLD_START : 0u;  // one collision = one manifold
            c.reserved = 0;
            out[slot] = c;
        }
        slot += 1;
    }
}
__global__ __launch_bounds__(256) void k_svc_emit(SvcEmitArgs a) { svc_emit_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_svc_emit_many, SvcEmitArgs, svc_emit_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_svc_emit, k_svc_emit_many, SvcEmitArgs, 256)
static_assert(sizeof(SvcCountArgs) % 8 == 0 && sizeof(SvcScanArgs) % 8 == 0 && sizeof(SvcEmitArgs) % 8 == 0, "argument blocks travel as 8-byte words");

}  // namespace

static const int s_svc_many_registered = (ivx_many_register(IVX_MK_SVC_COUNT, many_svc_count, sizeof(SvcCountArgs)),
                                          ivx_many_register(IVX_MK_SVC_SCAN, many_svc_scan, sizeof(SvcScanArgs)),
                                          ivx_many_register(IVX_MK_SVC_EMIT, many_svc_emit, sizeof(SvcEmitArgs)), 0);

int ivx_launch_sphere_contacts(ivx_grid* g, const uint32_t lo[3], const uint32_t cc[3], const int32_t vlo[3], const int32_t vhi[3],
                               const float rotation_xyzw[4], const float translation[3], const float center[3], const float seg_vec[3], float radius,
                               uint64_t id_a, uint64_t id_b, uint32_t body_a, uint32_t body_b, const float response[3], uint32_t* d_counts,
                               uint32_t* d_offsets, uint32_t* d_total, ivx_contact* d_out, uint32_t cap, int emit, int mode) {
    SvcParams p;
    p.g = ivx_view(g);
    p.mode = (uint32_t)mode;
    for (int d = 0; d < 3; ++d) p.v[d] = mode == 2 ? seg_vec[d] : 0.0f;
    p.len2 = (p.v[0] * p.v[0] + p.v[1] * p.v[1]) + p.v[2] * p.v[2];
    for (int d = 0; d < 3; ++d) {
        p.lo[d] = lo[d];
        p.cc[d] = cc[d];
        p.vlo[d] = vlo[d];
        p.vhi[d] = vhi[d];
        p.t[d] = translation[d];
        p.c[d] = center[d];
    }
    p.q_inv = Q4{-rotation_xyzw[0], -rotation_xyzw[1], -rotation_xyzw[2], rotation_xyzw[3]};
    p.r = radius;
    p.extent = g->extent;
    {  // ContactID::from_two_u64_and_n_indices: the part that does not depend on the voxel
        auto mix = [](uint64_t state) {
            state += 0x9E3779B97F4A7C15ull;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        p.id_ab = mix(id_a ^ mix(id_b));
    }
    p.body_a = body_a;
    p.body_b = body_b;
    p.restitution = response[0];
    p.static_friction = response[1];
    p.dynamic_friction = response[2];
    const uint32_t n_box = cc[0] * cc[1] * cc[2];
    if (!emit) {
        SvcCountArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.p = p, ca.flags = g->flags, ca.counts = d_counts;
        if (!ivx_many_try(g->ctx, g, IVX_MK_SVC_COUNT, n_box, ca)) IVX_KLAUNCH(k_svc_count, dim3(n_box), dim3(256), 0, g->ctx->stream, ca);
        SvcScanArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.counts = d_counts, sa.offsets = d_offsets, sa.total = d_total, sa.n = n_box;
        if (!ivx_many_try(g->ctx, g, IVX_MK_SVC_SCAN, 1u, sa)) IVX_KLAUNCH(k_svc_scan, dim3(1), dim3(256), 0, g->ctx->stream, sa);
    } else {
        SvcEmitArgs ea;
        memset(&ea, 0, sizeof(ea));
        ea.p = p, ea.flags = g->flags, ea.offsets = d_offsets, ea.out = d_out, ea.cap = cap;
        if (!ivx_many_try(g->ctx, g, IVX_MK_SVC_EMIT, n_box, ea)) IVX_KLAUNCH(k_svc_emit, dim3(n_box), dim3(256), 0, g->ctx->stream, ea);
    }
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
