// a3 — SDF sample: atomic SDF graph -> quantised voxels + chunk classification, one workgroup per
// 16^3 chunk, one thread per (i,j) row of 16 voxels along k.
//
// Reference behaviour reproduced (engine/crates/impact_voxel/src/):
//   SDFVoxelGenerator::generate_chunk                 generation.rs:293-371
//   SDFGenerator::compute_signed_distances_for_block  generation/sdf/atomic.rs:633-875
//   update_signed_distances_for_block[_packed]        generation/sdf/atomic.rs:1601-1658
//   block test positions                              generation/sdf/atomic.rs:1661-1797
//   primitives / smooth ops                           atomic.rs:1183-1291, generation/sdf.rs:47-102
//   VoxelSignedDistance::from_f32                     lib.rs:197-201
//   VoxelChunk::create_for_generated_voxels           object.rs:1890-1964
//
// Mapping to CDNA4: the reference's per-block "stack machine" of 4096-float arrays lives in LDS as
// stack[level][k][thread] (bank-conflict free: consecutive lanes hit consecutive banks); each thread
// only ever touches its own 16-voxel row, so the only barriers are around the 14 block test positions
// of a combination node. Positions advance along k by repeated `pos += dz` exactly like the reference
// loop, which keeps results bit-identical (no FMA: built with -ffp-contract=off; IEEE sqrt/div).
// The row is written back as one 16-byte store per plane per thread (4 KiB contiguous per workgroup).
// HBM traffic: 2 B/voxel written (sdf + type), nothing read.
#include "ivx_internal.hpp"
#include "table_roles.hpp"

namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// Correctly rounded sqrt for the magnitudes this kernel meets (squared distances in voxel units: zero or well inside the normal
// range). Same algorithm as the compiler's IEEE lowering of sqrtf — hardware estimate, then pick the neighbour float whose
// residual changes sign — without its rescaling of tiny inputs and its inf/NaN class test: 9 instructions instead of 16, and
// the evaluator is bound by VALU issue. sqrt(0) = 0 falls out (the "one ulp down" candidate is a NaN and never wins).
__device__ __forceinline__ float sqrt_rn(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float r = r_dn <= 0.0f ? s_dn : s;
    r = r_up > 0.0f ? s_up : r;
    return r;
}
__device__ __forceinline__ float len3(V3 a) { return sqrt_rn(dot3(a, a)); }
// f32::min / f32::max of the reference. No NaN reaches these (distances are finite), and which zero comes out of
// min(+0, -0) cannot reach an output (a distance of either zero quantises to 0 and compares alike), so the hardware
// instructions stand in for compare + select: one VALU op instead of two in a kernel bound by VALU issue. (Inline asm:
// through fminf the compiler would first canonicalise operands it cannot prove quiet, which costs the instruction back.)
__device__ __forceinline__ float min_rs(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_rs(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ bool sneg(float f) { return (__float_as_uint(f) >> 31) != 0; }
__device__ __forceinline__ uint32_t negmask(V3 a) { return (sneg(a.x) ? 1u : 0u) | (sneg(a.y) ? 2u : 0u) | (sneg(a.z) ? 4u : 0u); }

struct Box {
    V3 lo, hi;
};
// impact_geometry/src/axis_aligned_box.rs:253-268
__device__ __forceinline__ bool contains_box(Box self, Box o) { return (negmask(sub(o.lo, self.lo)) | negmask(sub(self.hi, o.hi))) == 0; }
__device__ __forceinline__ bool lies_outside(Box self, Box o) { return (negmask(sub(o.hi, self.lo)) | negmask(sub(self.hi, o.lo))) != 0; }

// glam Mat4::transform_point3a order: ((c0*x + c1*y) + c2*z) + c3
__device__ __forceinline__ V3 xform_point(const float* m, V3 p) {
    V3 r = scale(mk(m[0], m[1], m[2]), p.x);
    r = add(scale(mk(m[4], m[5], m[6]), p.y), r);
    r = add(scale(mk(m[8], m[9], m[10]), p.z), r);
    r = add(mk(m[12], m[13], m[14]), r);
    return r;
}
// axis_aligned_box.rs:350-366
__device__ __forceinline__ Box aabb_of_transformed(Box b, const float* m) {
    V3 c = xform_point(m, scale(add(b.lo, b.hi), 0.5f));
    V3 h = scale(sub(b.hi, b.lo), 0.5f);
    V3 a0 = mk(fabsf(m[0]), fabsf(m[1]), fabsf(m[2]));
    V3 a1 = mk(fabsf(m[4]), fabsf(m[5]), fabsf(m[6]));
    V3 a2 = mk(fabsf(m[8]), fabsf(m[9]), fabsf(m[10]));
    V3 he = add(add(scale(a0, h.x), scale(a1, h.y)), scale(a2, h.z));
    return {sub(c, he), add(c, he)};
}

// generation/sdf.rs:89-92
__device__ __forceinline__ float smooth_union(float d1, float d2, float s, float q) {
    float h = max_rs(s - fabsf(d1 - d2), 0.0f);
    return min_rs(d1, d2) - (h * h) * q;
}
__device__ __forceinline__ float combine(uint32_t kind, float a, float b, float s, float q) {
    if (kind == 7u) return s == 0.0f ? min_rs(a, b) : smooth_union(a, b, s, q);
    if (kind == 8u) return s == 0.0f ? max_rs(a, -b) : -smooth_union(-a, b, s, q);
    return s == 0.0f ? max_rs(a, b) : -smooth_union(-a, -b, s, q);
}


// Element-wise application of one combination over the thread's 16 voxels, specialised so that no
// branch on the (workgroup-uniform) operator kind / smoothness sits inside the unrolled loop.
template <int KIND, bool SMOOTH>
__device__ __forceinline__ float combine_t(float a, float b, float s, float q) {
    if (KIND == 7) return SMOOTH ? smooth_union(a, b, s, q) : min_rs(a, b);
    if (KIND == 8) return SMOOTH ? -smooth_union(-a, b, s, q) : max_rs(a, -b);
    return SMOOTH ? -smooth_union(-a, -b, s, q) : max_rs(a, b);
}
// Element k of a thread's column of a stack level: LDS at d[k * 256] — except, in the trimmed launch of the two-level class
// (k_sdf_eval<1>), row 15 of the second dense level, which every thread keeps in a register (`r15`; `t` says that this level is
// that one, wave-uniform). The 768 bytes this saves bring a workgroup's stack from 32 768 to 32 000 bytes, and five of them fit a CU
// instead of four (LDS is handed out in 1 280-byte granules on this part: measured, sample stage 0.117 -> 0.108 ms).
#define IVX_LV_GET(d, k, t, r15) (((k) == 15 && (t)) ? (r15) : (d)[(k) * 256])
#define IVX_LV_SET(d, k, t, r15, val) \
    do {                               \
        const float lv_val_ = (val);   \
        if ((k) == 15 && (t)) (r15) = lv_val_; \
        else (d)[(k) * 256] = lv_val_; \
    } while (0)

template <int KIND, bool SMOOTH>
__device__ __forceinline__ void apply_rows(float* d1, bool t1, const float* d2, bool t2, float& r15, bool c1, bool c2, float v1, float v2, float s, float q) {
    if (c1) {
#pragma unroll
        for (int k = 0; k < 16; ++k) IVX_LV_SET(d1, k, t1, r15, (combine_t<KIND, SMOOTH>(v1, IVX_LV_GET(d2, k, t2, r15), s, q)));
    } else if (c2) {
#pragma unroll
        for (int k = 0; k < 16; ++k) IVX_LV_SET(d1, k, t1, r15, (combine_t<KIND, SMOOTH>(IVX_LV_GET(d1, k, t1, r15), v2, s, q)));
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) IVX_LV_SET(d1, k, t1, r15, (combine_t<KIND, SMOOTH>(IVX_LV_GET(d1, k, t1, r15), IVX_LV_GET(d2, k, t2, r15), s, q)));
    }
}
__device__ __forceinline__ void apply_rows_dispatch(uint32_t kind, float* d1, bool t1, const float* d2, bool t2, float& r15, bool c1, bool c2, float v1, float v2,
                                                    float s, float q) {
    const bool smooth = s != 0.0f;
    if (kind == 7u) {
        if (smooth) apply_rows<7, true>(d1, t1, d2, t2, r15, c1, c2, v1, v2, s, q);
        else apply_rows<7, false>(d1, t1, d2, t2, r15, c1, c2, v1, v2, s, q);
    } else if (kind == 8u) {
        if (smooth) apply_rows<8, true>(d1, t1, d2, t2, r15, c1, c2, v1, v2, s, q);
        else apply_rows<8, false>(d1, t1, d2, t2, r15, c1, c2, v1, v2, s, q);
    } else {
        if (smooth) apply_rows<9, true>(d1, t1, d2, t2, r15, c1, c2, v1, v2, s, q);
        else apply_rows<9, false>(d1, t1, d2, t2, r15, c1, c2, v1, v2, s, q);
    }
}

// lib.rs:197-201: (v * 50.0) as i8 — truncate toward zero, saturate, NaN -> 0
// (v_cvt_i32_f32 is that cast onto i32 — toward zero, saturating, NaN -> 0 — and the clamp to the i8 range comes out the same from
// there: three instructions per voxel where the comparisons were eight, in the one kernel that is bound by VALU issue)
__device__ __forceinline__ int sd_from_f32(float v) {
    const float s = v * 50.0f;
    int i;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(s));
    return min(max(i, -128), 127);
}

// developer trace probes (make TRACE=1): the evaluator's by default; `make TRACE=1 EXTRA=-DIVX_TRACE_PREPASS` traces the pre-pass
// instead (both kernels run in the sample stage and share the trace buffer)
#if defined(IVX_WG_TRACE) && defined(IVX_TRACE_PREPASS)
#define IVX_TP(g, entry, slot) IVX_T(g, entry, slot)
#define IVX_TE(g, entry, slot) \
    do {                       \
    } while (0)
#else
#define IVX_TP(g, entry, slot) \
    do {                       \
    } while (0)
#define IVX_TE(g, entry, slot) IVX_T(g, entry, slot)
#endif

struct SampleParams {
#ifdef IVX_WG_TRACE
    unsigned long long* trace;
#endif
    uint32_t cx, cy, cz, x_off;
    uint32_t shape[3];
    float shifted_center[3];
    uint32_t n_nodes, stack_size;
    uint32_t voxel_type;
};

// the low bytes of 16 ints as 16 bytes: three byte permutes per word (v_perm_b32 selector bytes: 0-3 = second operand's, 4-7 = first's)
__device__ __forceinline__ uint4 pack16(const int* v) {
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t lo = __builtin_amdgcn_perm((uint32_t)v[4 * q + 1], (uint32_t)v[4 * q], 0x0C0C0400u);      // bytes: v0.b0, v1.b0, 0, 0
        const uint32_t hi = __builtin_amdgcn_perm((uint32_t)v[4 * q + 3], (uint32_t)v[4 * q + 2], 0x04000C0Cu);  // bytes: 0, 0, v2.b0, v3.b0
        w[q] = lo | hi;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Tail shared by sampling and classification (generation.rs:327-371 + object.rs:1890-1964): given the
// thread's 16 quantised distances (and types), classify the chunk, canonicalise void chunks to
// maximally-outside voxels (the reference drops their data, object/sdf.rs:486-489) and store.
__device__ __forceinline__ void classify_and_store(int* sd, uint4 types, bool types_uniform_in, uint8_t first_type, int8_t* sdf_out,
                                                   uint8_t* type_out, ivx_chunk_info* info_out, uint32_t chunk, uint32_t tid,
                                                   bool set_type, uint32_t voxel_type, bool compact, uint32_t* s_votes,
                                                   uint16_t* signs_out = nullptr, uint8_t* kface_out = nullptr) {
    // the three per-thread predicates from the smallest and the largest of the 16 distances (v_min3 / v_max3: 16 instructions where
    // the per-voxel comparisons were ~100)
    int lo = sd[0], hi = sd[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        lo = min(lo, sd[k]);
        hi = max(hi, sd[k]);
    }
    const bool any_nonempty = lo < 0, any_nonvoid = lo <= SD_VOID_LIMIT, all_inside = hi == -128;
    // the three workgroup votes through four words of LDS the caller lends (`s_votes`: in k_sdf_eval the tail of the stack, dead by now —
    // the kernel keeps no LDS of its own, so that five 32 KB stacks fit a CU), one barrier pair instead of three
    const uint32_t mine = (__ballot(any_nonempty) ? 1u : 0u) | (__ballot(any_nonvoid) ? 2u : 0u) | (__ballot(!(all_inside && types_uniform_in)) ? 4u : 0u);
    __syncthreads();  // every wave is past its last use of the memory the votes go to
    if ((tid & 63u) == 0u) s_votes[tid >> 6] = mine;
    __syncthreads();
    const uint32_t votes = (s_votes[0] | s_votes[1]) | (s_votes[2] | s_votes[3]);
    const int only_empty = !(votes & 1u), is_void = !(votes & 2u), uniform = !(votes & 4u);
    uint32_t kind = is_void ? KIND_VOID : ((!only_empty && uniform) ? KIND_UNIFORM : KIND_NONUNIFORM);
    if (is_void) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sd[k] = 127;
        types = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    } else if (set_type) {
        // SameVoxelTypeGenerator: every voxel of a chunk with >=1 non-empty voxel gets the type,
        // all-empty chunks keep VoxelType::dummy() (generation.rs:348-365, voxel_type.rs:88-96)
        uint32_t t = only_empty ? 0xFFFFFFFFu : voxel_type * 0x01010101u;
        types = make_uint4(t, t, t, t);
        first_type = only_empty ? (uint8_t)TYPE_DUMMY : (uint8_t)voxel_type;
    }
    if (kind == KIND_NONUNIFORM || !compact) {  // Void / Uniform chunks are their 8-byte record (compact planes)
        size_t base = (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)tid * 16;
        const uint4 packed = pack16(sd);
        *reinterpret_cast<uint4*>(sdf_out + base) = packed;
        *reinterpret_cast<uint4*>(type_out + base) = types;
        if (signs_out && kind == KIND_NONUNIFORM) {
            // What the derive sweep and the mesher need of this chunk besides its planes, while the row is in registers: the row's 16-bit
            // "distance negative" mask (GridView::signs) and its bytes on the two k faces (GridView::kface). With them in place the sweep
            // reads no voxel plane at all — its own rows and its six neighbours' faces are 2 bytes per row (ivx_launch_derive, k_derive<true>).
            const uint32_t w[4] = {packed.x, packed.y, packed.z, packed.w};
            uint32_t m = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) m |= ((((w[q] >> 7) & 0x01010101u) * 0x00204081u >> 21) & 0xFu) << (4 * q);  // (row_mask of derive.hip)
            signs_out[(size_t)chunk * 256 + tid] = (uint16_t)m;
            uint8_t* kf = kface_out + (size_t)chunk * 1024 + tid;
            kf[0] = (uint8_t)(packed.x & 0xFFu);
            kf[256] = (uint8_t)(packed.w >> 24);
            kf[512] = (uint8_t)(types.x & 0xFFu);
            kf[768] = (uint8_t)(types.w >> 24);
        }
    }
    if (tid == 0) {
        ivx_chunk_info ci;
        ci.kind = (uint8_t)kind;
        ci.gen_kind = (uint8_t)kind;
        ci.flags = (kind == KIND_NONUNIFORM && only_empty) ? (uint8_t)CF_ONLY_EMPTY : (uint8_t)0;
        ci.uniform_type = kind == KIND_UNIFORM ? first_type : (uint8_t)0;
        ci.face_dist = 0;
        ci.region_count = 0;
        ci.boundary_region_count = 0;
        info_out[chunk] = ci;
    }
}

// Block-level decision of one node (the reference's whole-block early-outs, atomic.rs:654-668,
// 788-806): leaves -> 0 evaluate, 1 constant +margin, 2 constant -margin; combinations -> 0 apply,
// 1 node domain lies outside the block (apply only if a test position fails).
__device__ __forceinline__ uint32_t node_mode(const ivx_sdf_processed_node* nd, Box block) {
    const uint32_t kind = nd->kind;
    if (kind > 2u && kind < 7u) return 0u;
    const Box bn = aabb_of_transformed(block, nd->transform);
    const Box dom{mk(nd->domain_lo[0], nd->domain_lo[1], nd->domain_lo[2]), mk(nd->domain_hi[0], nd->domain_hi[1], nd->domain_hi[2])};
    const bool outside = lies_outside(dom, bn);
    if (kind >= 7u) return outside ? 1u : 0u;
    if (outside) return 1u;
    const float margin = nd->margin;
    V3 ih;
    if (kind == 0u) {
        float e = nd->a * 0.57735026f + (-margin);
        ih = mk(e, e, e);
    } else if (kind == 1u) {
        float e = nd->b * 0.57735026f + (-margin);
        ih = mk(e, e + nd->a, e);
    } else {
        ih = mk(nd->a + (-margin), nd->b + (-margin), nd->c + (-margin));
    }
    const Box interior{mk(-ih.x, -ih.y, -ih.z), ih};
    return contains_box(interior, bn) ? 2u : 0u;
}


// Per-chunk scalar pre-pass (one THREAD per chunk): runs the node program on a conservative INTERVAL
// [lo, hi] of each stack level over the chunk, with exact tracking of block constants.
//   * leaf in a fill mode (the reference's whole-block early-outs): exact constant +-margin;
//   * evaluated leaf: distance bounds from the node-space AABB of the chunk, widened by a rounding slack;
//   * every combination is monotone in its operands (union/intersection increasing in both,
//     subtraction increasing in the first and decreasing in the second; smooth forms have partial
//     derivatives in [0,1]), so bounds propagate through the same f32 formulas;
//   * a combination whose apply decision needs per-voxel test values takes the hull of both outcomes.
// Outcome per chunk: an exact constant (chunk finished), "every voxel quantises to +127" (root lo >=
// 2.54 + one quantisation step), "every voxel quantises to -128" (root hi <= -2.56 - one step), or NaN =
// evaluate per voxel. The saturating quantisation (lib.rs:197-201) makes the two bound cases exact.
__device__ __forceinline__ float slack(float v) { return 1e-3f + 1e-5f * fabsf(v); }

__device__ __forceinline__ void leaf_bounds(const ivx_sdf_processed_node* nd, Box bn, float& lo, float& hi) {
    const uint32_t kind = nd->kind;
    V3 blo = bn.lo, bhi = bn.hi;
    if (kind == 1u) {  // capsule: y -= clamp(y, -h, h) is monotone in y
        const float h = nd->a;
        float cl = blo.y < -h ? -h : (blo.y > h ? h : blo.y), ch = bhi.y < -h ? -h : (bhi.y > h ? h : bhi.y);
        blo.y -= cl;
        bhi.y -= ch;
    }
    // per-component bounds of |p|
    const V3 amin = mk((blo.x <= 0.0f && bhi.x >= 0.0f) ? 0.0f : fminf(fabsf(blo.x), fabsf(bhi.x)),
                       (blo.y <= 0.0f && bhi.y >= 0.0f) ? 0.0f : fminf(fabsf(blo.y), fabsf(bhi.y)),
                       (blo.z <= 0.0f && bhi.z >= 0.0f) ? 0.0f : fminf(fabsf(blo.z), fabsf(bhi.z)));
    const V3 amax = mk(fmaxf(fabsf(blo.x), fabsf(bhi.x)), fmaxf(fabsf(blo.y), fabsf(bhi.y)), fmaxf(fabsf(blo.z), fabsf(bhi.z)));
    if (kind == 2u) {
        const V3 qlo = mk(amin.x - nd->a, amin.y - nd->b, amin.z - nd->c), qhi = mk(amax.x - nd->a, amax.y - nd->b, amax.z - nd->c);
        lo = len3(mk(fmaxf(qlo.x, 0.0f), fmaxf(qlo.y, 0.0f), fmaxf(qlo.z, 0.0f))) + fminf(fmaxf(fmaxf(qlo.x, qlo.y), qlo.z), 0.0f);
        hi = len3(mk(fmaxf(qhi.x, 0.0f), fmaxf(qhi.y, 0.0f), fmaxf(qhi.z, 0.0f))) + fminf(fmaxf(fmaxf(qhi.x, qhi.y), qhi.z), 0.0f);
    } else {
        const float r = kind == 0u ? nd->a : nd->b;
        lo = len3(amin) - r;
        hi = len3(amax) - r;
    }
    lo -= slack(lo);
    hi += slack(hi);
}

// The node program is staged through LDS in tiles: per-node parameter fetches from HBM at ~1 us of
// dependent latency each were the whole cost of large programs (hundreds of nodes). Stride 33 dwords
// keeps lane-per-node reads conflict free.
constexpr int NODE_TILE = 64;
// Compact per-chunk program emitted by the pre-pass: op word = opcode<<28 | node kind<<24 | node index;
// OP_CONST carries the folded constant in the second word.
constexpr uint32_t OP_CAP = 128u, OP_OVERFLOW = 0xFFFFFFFFu;
constexpr uint32_t LONG_OPS = 10u;  // compact programs longer than this are evaluated first (see k_sdf_prepass's list appends)
constexpr uint32_t OP_CONST = 0u, OP_LEAF = 1u, OP_SCALE = 2u, OP_COMBINE = 3u, OP_COMBINE_OUTSIDE = 4u;
// OP_SKIP: the evaluator jumps `second word` ops ahead (itself included): the steps of a first operand that the pre-pass found to be out of
// its combination's reach stay in the stream behind one (the stream is append / truncate only), dead
constexpr uint32_t OP_SKIP = 5u;
constexpr int PRE_T = 64;      // chunks per pre-pass block: one per lane
constexpr int PRE_WAVES = 8;   // waves per pre-pass block: all of them take the nodes' box tests, the first walks the program
struct PaddedNode {
    ivx_sdf_processed_node n;
    uint32_t pad;
};
static_assert(sizeof(PaddedNode) == 132, "node tile stride");
__device__ __forceinline__ void load_node_tile(PaddedNode* tile, const ivx_sdf_processed_node* src, uint32_t count, uint32_t tid, uint32_t nthreads) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d = reinterpret_cast<uint32_t*>(tile);
    for (uint32_t i = tid; i < count * 32u; i += nthreads) d[(i >> 5) * 33u + (i & 31u)] = s[i];
}

// Super-blocks of 4 x 4 x 4 chunks: bit n of a super-block's mask says that node n takes its "domain lies outside the block"
// early-out (atomic.rs:654-668, 788-806) for EVERY chunk of the super-block. The block test is monotone — the node-space box
// of a chunk lies inside the node-space box of its super-block — so the bit follows from the same test on the super-block,
// taken with a safety distance that dwarfs the rounding of the two box computations. One thread per (super-block, node).
// The pre-pass then replaces whole far bodies by their folded constant and skips the box arithmetic of single far nodes: its
// cost no longer grows with the number of bodies in the scene.
constexpr int SUPER = 4;
constexpr uint32_t SUPER_MAX_WORDS = 64;  // programs of up to 2048 nodes get their super-block tables inside the pre-pass (k_sdf_super beyond)
// per super-block and node: .x = for a leaf, the root of the largest subtree starting at it whose nodes are all far (the
// pre-pass replaces the range by one constant), else the node's own index; .y = that subtree's folded constant
__device__ __forceinline__ bool range_all_far(const uint32_t* mask, uint32_t a, uint32_t b);
// the node takes its "domain lies outside the block" early-out for every chunk of the super-block `block`
__device__ __forceinline__ bool super_far(const ivx_sdf_processed_node* nd, Box block) {
    const uint32_t kind = nd->kind;
    if (kind > 2u && kind < 7u) return true;  // translation / rotation / scaling have no test of their own
    const Box bn = aabb_of_transformed(block, nd->transform);
    const float eps = 1e-2f;
    return bn.hi.x - nd->domain_lo[0] < -eps || bn.hi.y - nd->domain_lo[1] < -eps || bn.hi.z - nd->domain_lo[2] < -eps ||
           nd->domain_hi[0] - bn.lo.x < -eps || nd->domain_hi[1] - bn.lo.y < -eps || nd->domain_hi[2] - bn.lo.z < -eps;
}
// skip-table entry of node n from the super-block's far bits `mask`
__device__ __forceinline__ uint2 super_skip_of(const ivx_sdf_processed_node* nodes, const uint32_t* mask, uint32_t n) {
    uint32_t target = n, fbits = 0;
    if (nodes[n].kind <= 2u) {
        uint32_t r = nodes[n].reserved[1];  // outermost subtree starting at this leaf, then inwards
        while (r != n) {
            if (range_all_far(mask, n, r)) {
                target = r;
                fbits = nodes[r].reserved[0];
                break;
            }
            r = nodes[r].reserved[2];
        }
    }
    return make_uint2(target, fbits);
}
__global__ __launch_bounds__(64) void k_sdf_super(SampleParams p, const ivx_sdf_processed_node* __restrict__ nodes, uint32_t* __restrict__ super_mask,
                                                  uint2* __restrict__ super_skip, uint32_t words, uint32_t sy, uint32_t sz, ivx_roles::PresetArgs preset) {
    extern __shared__ uint32_t s_mask[];  // [words]
    const uint32_t sb = blockIdx.x, lane = threadIdx.x;
    ivx_roles::role_preset(preset, sb * 64u + lane);  // the step's first kernel also presets the scratch words of the step's stages
    const uint32_t sk = sb % sz, sj = (sb / sz) % sy, si = sb / (sz * sy);
    const V3 lo = sub(mk((float)((si * SUPER + p.x_off) * 16u), (float)(sj * SUPER * 16u), (float)(sk * SUPER * 16u)),
                      mk(p.shifted_center[0], p.shifted_center[1], p.shifted_center[2]));
    const Box block{lo, add(lo, mk(16.0f * SUPER, 16.0f * SUPER, 16.0f * SUPER))};
    for (uint32_t n0 = 0; n0 < words * 32u; n0 += 64u) {
        const uint32_t n = n0 + lane;
        bool far = false;
        if (n < p.n_nodes) far = super_far(nodes + n, block);
        const unsigned long long b = __ballot(far);
        if (lane == 0 && (n0 >> 5) < words) s_mask[n0 >> 5] = (uint32_t)b;
        if (lane == 32 && (n0 >> 5) + 1u < words) s_mask[(n0 >> 5) + 1u] = (uint32_t)(b >> 32);
    }
    __syncthreads();
    for (uint32_t w = lane; w < words; w += 64u) super_mask[(size_t)sb * words + w] = s_mask[w];
    for (uint32_t n = lane; n < p.n_nodes; n += 64u) super_skip[(size_t)sb * words * 32u + n] = super_skip_of(nodes, s_mask, n);
}

// every node of [a, b] far?
__device__ __forceinline__ bool range_all_far(const uint32_t* mask, uint32_t a, uint32_t b) {
    for (uint32_t w = a >> 5; w <= (b >> 5); ++w) {
        const uint32_t lo = w == (a >> 5) ? (a & 31u) : 0u, hi = w == (b >> 5) ? (b & 31u) : 31u;
        const uint32_t m = (hi == 31u ? 0xFFFFFFFFu : ((1u << (hi + 1u)) - 1u)) & ~((1u << lo) - 1u);
        if ((mask[w] & m) != m) return false;
    }
    return true;
}

// `ahead`: the launch runs a step ahead of its sample stage (ivx_grid_set_sample_ahead): `info_out` is the grid's shadow record array, where
// a chunk that is NOT settled here gets a record of kind AHEAD_OPEN, so that the evaluator launch that commits the shadow knows which are
constexpr uint8_t AHEAD_OPEN = 0xFFu;
__global__ __launch_bounds__(PRE_T * PRE_WAVES) void k_sdf_prepass(SampleParams p, const ivx_sdf_processed_node* __restrict__ nodes,
                                                       uint32_t* __restrict__ prog_len,
                                                       uint2* __restrict__ prog_ops, uint32_t* __restrict__ eval_count,
                                                       uint32_t* __restrict__ eval_list, uint32_t list_stride, ivx_chunk_info* __restrict__ info_out,
                                                       const uint32_t* __restrict__ super_mask, const uint2* __restrict__ super_skip, uint32_t words,
                                                       uint32_t sy, uint32_t sz, uint32_t n_sb, uint32_t fused_super, uint32_t ahead, ivx_roles::PresetArgs preset) {
    // (with `fused_super` this is the step's first kernel and hosts the presets of the later stages' scratch words; the sampler's own
    // counters are never among them: other blocks of this launch are adding to those)
    ivx_roles::role_preset(preset, blockIdx.x * (uint32_t)(PRE_T * PRE_WAVES) + threadIdx.x);
    __shared__ uint32_t s_mask_all[SUPER_MAX_WORDS];  // fused_super: the super-block's far bit of every node of the program
    __shared__ uint2 s_skip[NODE_TILE];
    __shared__ uint32_t s_far[NODE_TILE / 32];
    __shared__ float s_lo[16][PRE_T];
    __shared__ float s_hi[16][PRE_T];
    __shared__ uint16_t s_start[16][PRE_T];  // op-stream position where each stack level's steps begin
    __shared__ uint8_t s_need[16][PRE_T];    // LDS levels the evaluator needs to produce this stack level (0 for a constant)
    __shared__ PaddedNode s_nodes[NODE_TILE];
    // What a node's step needs from the node's and the chunk's geometry alone — the block test (node_mode) and an evaluated leaf's
    // distance bounds — does not depend on the stack: the block's waves work it out for a tile of nodes side by side (wave w takes
    // nodes w, w + PRE_WAVES, ...), then the first wave walks the program with these tables. One wave doing both spent most of the kernel,
    // 33 nodes one after the other at one wave per SIMD, in this arithmetic.
    __shared__ uint8_t t_mode[NODE_TILE][PRE_T];  // leaf: 0 evaluate, 1 / 2 constant +-margin; combination: 1 = must apply
    __shared__ float t_lo[NODE_TILE][PRE_T];
    __shared__ float t_hi[NODE_TILE][PRE_T];
    const uint32_t tid = threadIdx.x & (PRE_T - 1u);  // lane = chunk of the super-block
    const uint32_t wv = threadIdx.x / PRE_T;
    // one block per super-block of 4 x 4 x 4 chunks (so its threads share the super-block's far bits and skip table, staged in
    // LDS with the node tile); threads whose chunk lies beyond the grid still take part in the tile loads and barriers: they
    // redo a chunk of the grid and write nothing
    static_assert(PRE_T == SUPER * SUPER * SUPER, "one thread per chunk of a super-block");
    // (a launch of fewer blocks than super-blocks walks them in turn: the pre-pass that runs a step ahead, beside another step's kernels,
    // keeps to a block per CU — two of these blocks fill a CU's LDS)
    for (uint32_t sb = blockIdx.x; sb < n_sb; sb += gridDim.x) {
    if (sb != blockIdx.x) __syncthreads();
    const uint32_t sk = sb % sz, sj = (sb / sz) % sy, si = sb / (sz * sy);
    const uint32_t ci_raw = si * SUPER + (tid >> 4), cj_raw = sj * SUPER + ((tid >> 2) & 3u), ck_raw = sk * SUPER + (tid & 3u);
    const bool mine = ci_raw < p.cx && cj_raw < p.cy && ck_raw < p.cz;
    const uint32_t ci = min(ci_raw, p.cx - 1u), cj = min(cj_raw, p.cy - 1u), ck = min(ck_raw, p.cz - 1u);
    const uint32_t chunk = (ci * p.cy + cj) * p.cz + ck;
    const uint32_t oi = (ci + p.x_off) * 16u, oj = cj * 16u, ok = ck * 16u;
    const V3 origin_root = sub(mk((float)oi, (float)oj, (float)ok), mk(p.shifted_center[0], p.shifted_center[1], p.shifted_center[2]));
    const Box block{origin_root, add(origin_root, mk(16.0f, 16.0f, 16.0f))};
    // (the chunk's voxels themselves lie at origin + 0 .. 15: what an evaluated leaf's distance bounds are taken over — the reference's block
    // tests use the 16-box above)
    const Box vox_block{origin_root, add(origin_root, mk(15.0f, 15.0f, 15.0f))};
    if (fused_super) {
        // k_sdf_super's far test for this block's own super-block, all nodes, 64 per wave and round (the tables are a function of the
        // program and the grid; building them here instead of in a launch of their own costs the block ~2 us and the step one launch less)
        const V3 slo = sub(mk((float)((si * SUPER + p.x_off) * 16u), (float)(sj * SUPER * 16u), (float)(sk * SUPER * 16u)),
                           mk(p.shifted_center[0], p.shifted_center[1], p.shifted_center[2]));
        const Box sblock{slo, add(slo, mk(16.0f * SUPER, 16.0f * SUPER, 16.0f * SUPER))};
        for (uint32_t n0 = wv * 64u; n0 < words * 32u; n0 += 64u * PRE_WAVES) {
            const uint32_t n = n0 + tid;
            bool far = false;
            if (n < p.n_nodes) far = super_far(nodes + n, sblock);
            const unsigned long long b = __ballot(far);
            if (tid == 0 && (n0 >> 5) < words) s_mask_all[n0 >> 5] = (uint32_t)b;
            if (tid == 32 && (n0 >> 5) + 1u < words) s_mask_all[(n0 >> 5) + 1u] = (uint32_t)(b >> 32);
        }
        __syncthreads();
    }
    uint32_t top = 0, cmask = 0;  // cmask bit: level is an EXACT block constant (lo == hi == value)
    uint32_t bare = 0;            // bare bit: the level's live steps are one evaluated leaf (what the evaluator combines from registers)
    bool class_dropped = false, poisoned = false;  // (saturation classes, see the combination step)
    // Compact program of this chunk: steps of constant sub-expressions collapse into one OP_CONST (the
    // steps of a stack level are a contiguous tail of the stream, so folding = truncate + re-emit).
    uint2* ops = prog_ops + (size_t)chunk * OP_CAP;
    uint32_t pos = 0;
    auto emit = [&](uint32_t w, uint32_t v) {
        if (pos < OP_CAP) ops[pos] = make_uint2(w, v);
        pos += 1;
    };
    int skip_until = -1;  // nodes up to here belong to a subtree that was replaced by its far constant
    IVX_TP(p, sb, 0);
    for (uint32_t n = 0; n < p.n_nodes; ++n) {
        if ((n % NODE_TILE) == 0u) {  // stage the next tile of the node program (and of the super-block's tables) in LDS
            __syncthreads();
            const uint32_t tile_cnt = min((uint32_t)NODE_TILE, p.n_nodes - n);
            load_node_tile(s_nodes, nodes + n, tile_cnt, threadIdx.x, PRE_T * PRE_WAVES);
            if (wv == 0u) {
                if (fused_super) {
                    if (n + tid < p.n_nodes) s_skip[tid] = super_skip_of(nodes, s_mask_all, n + tid);
                    if (tid < NODE_TILE / 32 && (n >> 5) + tid < words) s_far[tid] = s_mask_all[(n >> 5) + tid];
                } else {
                    if (n + tid < p.n_nodes) s_skip[tid] = super_skip[(size_t)sb * words * 32u + n + tid];
                    if (tid < NODE_TILE / 32 && (n >> 5) + tid < words) s_far[tid] = super_mask[(size_t)sb * words + (n >> 5) + tid];
                }
            }
            __syncthreads();
            for (uint32_t q = wv; q < tile_cnt; q += PRE_WAVES) {
                const ivx_sdf_processed_node* tn = &s_nodes[q].n;
                const uint32_t tk = tn->kind;
                const bool tfar = (s_far[q >> 5] >> (q & 31u)) & 1u;
                uint32_t mode = 0u;
                float blo = 0.0f, bhi = 0.0f;
                if (tk <= 2u) {
                    mode = tfar ? 1u : node_mode(tn, block);
                    if (mode == 0u) leaf_bounds(tn, aabb_of_transformed(vox_block, tn->transform), blo, bhi);
                } else if (tk >= 7u) {
                    mode = (!tfar && node_mode(tn, block) == 0u) ? 1u : 0u;
                }
                t_mode[q][tid] = (uint8_t)mode;
                t_lo[q][tid] = blo;
                t_hi[q][tid] = bhi;
            }
            __syncthreads();
            if (n == 0u) IVX_TP(p, sb, 1);  // first tile staged, box tests done
        }
        if (wv != 0u) continue;  // (the other waves only meet the first at the tile boundaries)
        if ((int)n <= skip_until) continue;
        const ivx_sdf_processed_node* nd = &s_nodes[n % NODE_TILE].n;
        const uint32_t kind = nd->kind;
        if (kind <= 2u) {
            // the largest subtree that starts at this leaf and lies outside as a whole collapses to its folded constant
            const uint2 sk2 = s_skip[n % NODE_TILE];
            if (sk2.x != n) {
                const float v = __uint_as_float(sk2.y);
                s_start[top][tid] = (uint16_t)min(pos, 0xFFFFu);
                s_lo[top][tid] = v;
                s_hi[top][tid] = v;
                s_need[top][tid] = 0;
                cmask |= 1u << top;
                bare &= ~(1u << top);
                emit(OP_CONST << 28, __float_as_uint(v));
                top += 1;
                skip_until = (int)sk2.x;
                continue;
            }
            const uint32_t mode = t_mode[n % NODE_TILE][tid];
            s_start[top][tid] = (uint16_t)min(pos, 0xFFFFu);
            if (mode == 1u || mode == 2u) {
                const float v = mode == 1u ? nd->margin : -nd->margin;
                s_lo[top][tid] = v;
                s_hi[top][tid] = v;
                s_need[top][tid] = 0;
                cmask |= 1u << top;
                bare &= ~(1u << top);
                emit(OP_CONST << 28, __float_as_uint(v));
            } else {
                const float lo = t_lo[n % NODE_TILE][tid], hi = t_hi[n % NODE_TILE][tid];
                s_lo[top][tid] = lo;
                s_hi[top][tid] = hi;
                s_need[top][tid] = 1;
                cmask &= ~(1u << top);
                bare |= 1u << top;
                emit((OP_LEAF << 28) | (kind << 24) | n, 0u);
            }
            top += 1;
        } else if (kind == 5u) {
            s_lo[top - 1][tid] = s_lo[top - 1][tid] * nd->a;
            s_hi[top - 1][tid] = s_hi[top - 1][tid] * nd->a;
            bare &= ~(1u << (top - 1));
            if ((cmask >> (top - 1)) & 1u) {
                pos = s_start[top - 1][tid];
                emit(OP_CONST << 28, __float_as_uint(s_lo[top - 1][tid]));
            } else {
                emit((OP_SCALE << 28) | (kind << 24) | n, 0u);
            }
        } else if (kind >= 7u) {
            top -= 1;
            const bool c1 = (cmask >> (top - 1)) & 1u, c2 = (cmask >> top) & 1u;
            const float lo1 = s_lo[top - 1][tid], hi1 = s_hi[top - 1][tid], lo2 = s_lo[top][tid], hi2 = s_hi[top][tid];
            const float s = nd->a, q = nd->b;
            const bool must_apply = t_mode[n % NODE_TILE][tid] != 0u;
            const bool bare2 = (bare >> top) & 1u;
            if (c1 && c2) {
                const float r = combine(kind, lo1, lo2, s, q);
                if (must_apply || !(r >= nd->margin)) {
                    s_lo[top - 1][tid] = r;
                    s_hi[top - 1][tid] = r;
                }
                pos = s_start[top - 1][tid];
                s_need[top - 1][tid] = 0;
                bare &= ~(1u << (top - 1));
                emit(OP_CONST << 28, __float_as_uint(s_lo[top - 1][tid]));
            } else {
                // absorption: an exact constant operand that the other operand can never come within
                // the smoothing distance of is the result, exactly (h = 0 in generation/sdf.rs:89-92)
                bool absorbed = false;
                float av = 0.0f;
                const float slk = slack(lo1) + slack(lo2) + slack(hi1) + slack(hi2);
                if (kind == 7u) {
                    if (c1 && lo2 >= lo1 + s + slk) { absorbed = true; av = lo1; }
                    else if (c2 && must_apply && lo1 >= lo2 + s + slk) { absorbed = true; av = lo2; }
                } else if (kind == 9u) {
                    if (c1 && hi2 <= lo1 - s - slk) { absorbed = true; av = lo1; }
                    else if (c2 && must_apply && hi1 <= lo2 - s - slk) { absorbed = true; av = lo2; }
                } else {
                    if (c1 && lo2 >= -lo1 + s + slk) { absorbed = true; av = lo1; }
                    else if (c2 && must_apply && hi1 <= -lo2 - s - slk) { absorbed = true; av = -lo2; }
                }
                if (absorbed) {
                    s_lo[top - 1][tid] = av;
                    s_hi[top - 1][tid] = av;
                    s_need[top - 1][tid] = 0;
                    cmask |= 1u << (top - 1);
                    bare &= ~(1u << (top - 1));
                    pos = s_start[top - 1][tid];
                    emit(OP_CONST << 28, __float_as_uint(av));
                    continue;
                }
                // identity: a second operand — constant or not — that the first can never come within the smoothing distance of
                // leaves the first operand unchanged, voxel for voxel (min/max picks it and h = 0: x - 0 * q is x), whether or not the
                // reference applies the node — drop the operand and the combination from the chunk's program. (An evaluated leaf is
                // evaluated wherever the chunk meets its padded bounding box: around the box's corners it is tens of voxels away.)
                {
                    bool ident;
                    if (kind == 7u) ident = hi1 <= lo2 - s - slk;
                    else if (kind == 9u) ident = lo1 >= hi2 + s + slk;
                    else ident = lo1 >= -lo2 + s + slk;
                    if (ident) {
                        pos = s_start[top][tid];
                        continue;
                    }
                }
                // ... and its mirror image: a first operand that never comes within the smoothing distance of the second leaves the SECOND
                // operand, voxel for voxel, when the node is applied whatever its test voxels say (a node behind the test may also leave the first
                // standing; a subtraction would leave the second negated). The first operand's steps cannot be cut out of the stream: an
                // OP_SKIP over them takes the place of their first one.
                if (must_apply && kind != 8u && !c2) {
                    const bool mirror = kind == 7u ? lo1 >= hi2 + s + slk : hi1 <= lo2 - s - slk;
                    if (mirror) {
                        const uint32_t a = s_start[top - 1][tid], b = s_start[top][tid];
                        if (a < OP_CAP) ops[a] = make_uint2(OP_SKIP << 28, b - a);
                        s_lo[top - 1][tid] = lo2;
                        s_hi[top - 1][tid] = hi2;
                        s_need[top - 1][tid] = s_need[top][tid];
                        cmask &= ~(1u << (top - 1));
                        bare = (bare & ~(1u << (top - 1))) | ((bare2 ? 1u : 0u) << (top - 1));
                        continue;
                    }
                }
                // Saturation classes (ivx_sdf_annotate_host): the chunk's bytes are (v * 50) as i8 of the ROOT's value, so two values of a node are
                // interchangeable when they are equal or lie in the same saturated class of that node — and an operand that lies in one class
                // throughout the chunk goes when that leaves the other operand (`nd->c`: union — everywhere >= c; intersection — everywhere
                // <= c; subtraction — the second operand everywhere >= c). The lemmas hold for combinations that are applied; a node behind the
                // 14-position test reads its operands' values at the test voxels, which a dropped operand may have changed: a chunk whose
                // program comes to hold one AFTER such a drop is evaluated on the full program instead (`poisoned`).
                {
                    const float cth = nd->c;
                    const bool drop2 = kind == 9u ? hi2 <= cth : lo2 >= cth;
                    if (drop2) {
                        pos = s_start[top][tid];
                        class_dropped = true;
                        continue;
                    }
                    if (must_apply && kind != 8u && !c2) {
                        const bool drop1 = kind == 7u ? lo1 >= cth : hi1 <= cth;
                        if (drop1) {
                            const uint32_t a = s_start[top - 1][tid], b = s_start[top][tid];
                            if (a < OP_CAP) ops[a] = make_uint2(OP_SKIP << 28, b - a);
                            s_lo[top - 1][tid] = lo2;
                            s_hi[top - 1][tid] = hi2;
                            s_need[top - 1][tid] = s_need[top][tid];
                            cmask &= ~(1u << (top - 1));
                            bare = (bare & ~(1u << (top - 1))) | ((bare2 ? 1u : 0u) << (top - 1));
                            class_dropped = true;
                            continue;
                        }
                    }
                }
                if (!must_apply && class_dropped) poisoned = true;
                emit(((must_apply ? OP_COMBINE : OP_COMBINE_OUTSIDE) << 28) | (kind << 24) | n, 0u);
                {  // the first operand keeps its level (if it has one) while the second is evaluated; the result needs one — unless the second
                    // is one bare leaf and the combination unconditional: the evaluator then combines it from registers (eval_leaf_fused)
                    const uint32_t n1 = s_need[top - 1][tid], n2 = s_need[top][tid];
                    const bool fused = must_apply && !c2 && bare2;
                    const uint32_t nn = fused ? max(n1, 1u) : max(max(n1, (c1 ? 0u : 1u) + n2), 1u);
                    s_need[top - 1][tid] = (uint8_t)nn;
                }
                float rlo, rhi;
                if (kind == 8u) {  // subtraction: decreasing in the second operand
                    rlo = combine(kind, lo1, hi2, s, q);
                    rhi = combine(kind, hi1, lo2, s, q);
                } else {
                    rlo = combine(kind, lo1, lo2, s, q);
                    rhi = combine(kind, hi1, hi2, s, q);
                }
                rlo -= slack(rlo);
                rhi += slack(rhi);
                if (!must_apply) {  // may also stay operand 1
                    rlo = fminf(rlo, lo1);
                    rhi = fmaxf(rhi, hi1);
                }
                s_lo[top - 1][tid] = rlo;
                s_hi[top - 1][tid] = rhi;
                cmask &= ~(1u << (top - 1));
                bare &= ~(1u << (top - 1));
            }
        }
    }
    if (wv != 0u) continue;  // (no barrier below this line)
    IVX_TP(p, sb, 2);  // program walked
    const float lo = p.n_nodes ? s_lo[0][tid] : 1000.0f, hi = p.n_nodes ? s_hi[0][tid] : 1000.0f;  // no program: empty space
    float out = __uint_as_float(0x7FC00000u);  // NaN = evaluate per voxel
    if (p.n_nodes == 0u) cmask = 1u;
    if (cmask & 1u) out = lo;
    else if (lo >= 2.54f + 0.02f) out = 1000.0f;    // every voxel quantises to +127
    else if (hi <= -2.56f - 0.02f) out = -1000.0f;  // every voxel quantises to -128
    if (poisoned) out = __uint_as_float(0x7FC00000u), pos = OP_CAP + 1u;  // (evaluated per voxel, on the full program)
    if (oi >= p.shape[0] || oj >= p.shape[1] || ok >= p.shape[2]) out = 1000.0f;  // beyond the generator's grid: all +127
    bool to_fill = false;
    if (mine) {
        prog_len[chunk] = pos <= OP_CAP ? pos : OP_OVERFLOW;
        bool settled = false;
        if (out == out) {
            // A constant chunk is classified right here (create_for_generated_voxels, object.rs:1890-1964): every voxel
            // void (> +2.0) -> Void, every voxel maximally inside -> Uniform; such a chunk is its record and has no planes to
            // write (compact planes). Any other constant, or a chunk that straddles the generator's grid (voxels beyond it
            // are +127), is written out by k_sdf_fill.
            const int sdq = sd_from_f32(out);
            const bool whole = oi + 16u <= p.shape[0] && oj + 16u <= p.shape[1] && ok + 16u <= p.shape[2];
            const bool is_void = sdq > SD_VOID_LIMIT;  // voxels beyond the grid are void too
            const bool is_uniform = whole && sdq == -128;
            if (is_void || is_uniform) {
                ivx_chunk_info rec;
                rec.kind = rec.gen_kind = (uint8_t)(is_void ? KIND_VOID : KIND_UNIFORM);
                rec.flags = 0;
                rec.uniform_type = is_uniform ? (uint8_t)p.voxel_type : (uint8_t)0;
                rec.face_dist = 0;
                rec.region_count = 0;
                rec.boundary_region_count = 0;
                info_out[chunk] = rec;
                settled = true;
            } else {
                to_fill = true;
            }
        }
        if (ahead && !settled) {
            ivx_chunk_info rec;
            rec.kind = rec.gen_kind = AHEAD_OPEN;
            rec.flags = rec.uniform_type = 0;
            rec.face_dist = 0;
            rec.region_count = rec.boundary_region_count = 0;
            info_out[chunk] = rec;
        }
    }
    // chunks that need per-voxel evaluation go on a list for k_sdf_eval, constant chunks with planes to write on one for
    // k_sdf_fill (order is irrelevant); one atomic per wave and list (the block is one wave)
    static_assert(PRE_T == 64, "the list appends below assume one wave (the first of the block) doing them");
    {
        // Evaluation lists by the number of LDS levels the chunk's compact program needs (k_sdf_eval gets one launch per
        // class, with that much LDS: residency, hence throughput, is set by LDS): class 0: one level, 1: two (and the chunks whose
        // program did not fit OP_CAP, which run the full program, when that fits two), 2: more.
        // A constant chunk that still has planes to write (neither Void nor Uniform, or straddling the generator's grid) is
        // evaluated like the others: its compact program is the one folded constant (or, for the saturated-bound case, the
        // per-voxel program, which saturates to the same bytes) and the evaluator's store path handles the grid edge.
        const unsigned long long below = (1ull << tid) - 1ull;
        const bool ev = mine && (out != out || to_fill);
        const uint32_t need = pos <= OP_CAP ? (uint32_t)s_need[0][tid] : max(p.stack_size, 2u);  // (the full program may use every level, unfused)
        const uint32_t cls = need <= 1u ? 0u : (need == 2u ? 1u : 2u);
        // The first class (nearly every chunk of a smooth body) is listed longest program first: programs of more than LONG_OPS steps from
        // the front of its list, the others from the back. Workgroups start in list order, so the evaluator's last workgroups are short
        // ones and its tail — a tenth of the kernel with the chunks in arbitrary order — shrinks. Counters: [c] = entries of class c,
        // [3] = long entries of class 0, [4] = short entries of class 0.
        const bool is_long = pos > LONG_OPS;
#pragma unroll
        for (uint32_t c = 0; c < 3; ++c) {
            const unsigned long long be = __ballot(ev && cls == c);
            uint32_t base_e = 0;
            if (tid == 0 && be) base_e = atomicAdd(eval_count + c, (uint32_t)__popcll(be));
            base_e = __shfl(base_e, 0, 64);
            if (c == 0u) {
                const unsigned long long bl = __ballot(ev && cls == 0u && is_long), bs = be & ~bl;
                uint32_t base_l = 0, base_s = 0;
                if (tid == 0 && bl) base_l = atomicAdd(eval_count + 3, (uint32_t)__popcll(bl));
                if (tid == 0 && bs) base_s = atomicAdd(eval_count + 4, (uint32_t)__popcll(bs));
                base_l = __shfl(base_l, 0, 64);
                base_s = __shfl(base_s, 0, 64);
                if (ev && cls == 0u) {
                    if (is_long) eval_list[base_l + (uint32_t)__popcll(bl & below)] = chunk;
                    else eval_list[list_stride - 1u - (base_s + (uint32_t)__popcll(bs & below))] = chunk;
                }
            } else if (ev && cls == c) {
                eval_list[(size_t)c * list_stride + base_e + (uint32_t)__popcll(be & below)] = chunk;
            }
        }
    }
    IVX_TP(p, sb, 3);
    IVX_TP(p, sb, 4);
    IVX_TP(p, sb, 5);
    }
}

// ---- per-voxel evaluation ----------------------------------------------------------------------
// One stack-machine step each; shared by the compact per-chunk program (normal path) and the full
// node program (fallback when a chunk's compact program overflows OP_CAP).
// A box leaf under a transform that moves only z along the column: where the whole WAVE's columns lie inside the box's footprint in x and y
// (px = py = 0, i.e. cxy = +0 — the inside of a slab, a wall, a floor: most columns of a box-built scene) the length sqrt((0 + 0) + pz^2)
// is pz itself: 0 + t = t, and in binary floating point a correctly rounded square root of a correctly rounded square gives the number back
// when the square neither underflows nor overflows. Guard for that: a half extent of at least 2^-20 makes |z| - c either non-positive
// (pz = 0: sqrt(0) = 0) or at least 2^-45 (within a factor two of c the subtraction is exact and a multiple of c's last place; beyond, it is
// at least c/2), and coordinates below 2^40 keep the square finite. Eleven of the box's seventeen instructions per voxel, in a kernel bound by
// VALU issue. (The test is per wave: a divergent branch would run both loops.)
__device__ __forceinline__ bool box_column_inside(float cxy, float z0, float dzz, float half_z) {
    const bool ok = cxy == 0.0f && fabsf(z0) < 1.0e12f;
    return __all(ok ? 1 : 0) != 0 && half_z >= 9.5367431640625e-7f && fabsf(dzz) < 1.0e9f;
}

__device__ __forceinline__ void eval_leaf(const ivx_sdf_processed_node* nd, uint32_t kind, float* d, bool t15, float& r15, V3 origin_root, uint32_t ti, uint32_t tj) {
    const float* m = nd->transform;
    const V3 origin = xform_point(m, origin_root);
    const V3 dx = mk(m[0], m[1], m[2]), dy = mk(m[4], m[5], m[6]), dz = mk(m[8], m[9], m[10]);
    const V3 opx = add(origin, scale(dx, (float)ti));
    V3 pos = add(opx, scale(dy, (float)tj));
    const float pa = nd->a, pb = nd->b, pc = nd->c;
    // A transform without rotation or shear (translations and scalings: most nodes) moves only z along the column: `pos.x + 0` and
    // `pos.y + 0` are `pos.x` and `pos.y` again (a -0 may become +0, which no term below can tell apart), so every term in x and y alone
    // is the column's — and the sums keep the reference's order, ((x^2 + y^2) + z^2). Five of a sphere's nineteen instructions per voxel,
    // nine of a box's twenty-seven, in a kernel bound by VALU issue. (wave-uniform branch: the transform is the node's)
    if (dz.x == 0.0f && dz.y == 0.0f) {
        float z = pos.z;
        if (kind == 0u || kind == 1u) {
            float y = pos.y, r = pa;
            if (kind == 1u) {
                float c = y;
                if (c < -pa) c = -pa;
                if (c > pa) c = pa;
                y -= c;
                r = pb;
            }
            const float cxy = pos.x * pos.x + y * y;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                IVX_LV_SET(d, k, t15, r15, sqrt_rn(cxy + z * z) - r);
                z += dz.z;
            }
        } else {
            const float qx = fabsf(pos.x) - pa, qy = fabsf(pos.y) - pb;
            const float px = max_rs(qx, 0.0f), py = max_rs(qy, 0.0f);
            const float cxy = px * px + py * py, mxy = max_rs(qx, qy);
            if (box_column_inside(cxy, z, dz.z, pc)) {  // (wave-uniform) sqrt(0 + pz^2) is pz, see box_column_inside
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float qz = fabsf(z) - pc, pz = max_rs(qz, 0.0f);
                    IVX_LV_SET(d, k, t15, r15, pz + min_rs(max_rs(mxy, qz), 0.0f));
                    z += dz.z;
                }
                return;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float qz = fabsf(z) - pc, pz = max_rs(qz, 0.0f);
                IVX_LV_SET(d, k, t15, r15, sqrt_rn(cxy + pz * pz) + min_rs(max_rs(mxy, qz), 0.0f));
                z += dz.z;
            }
        }
        return;
    }
    if (kind == 0u) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            IVX_LV_SET(d, k, t15, r15, len3(pos) - pa);
            pos = add(pos, dz);
        }
    } else if (kind == 1u) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            V3 q = pos;
            float c = q.y;
            if (c < -pa) c = -pa;
            if (c > pa) c = pa;
            q.y -= c;
            IVX_LV_SET(d, k, t15, r15, len3(q) - pb);
            pos = add(pos, dz);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            V3 q = mk(fabsf(pos.x) - pa, fabsf(pos.y) - pb, fabsf(pos.z) - pc);
            V3 qp = mk(max_rs(q.x, 0.0f), max_rs(q.y, 0.0f), max_rs(q.z, 0.0f));
            IVX_LV_SET(d, k, t15, r15, len3(qp) + min_rs(max_rs(max_rs(q.x, q.y), q.z), 0.0f));
            pos = add(pos, dz);
        }
    }
}

// index of the LDS level that holds stack level `level`: the number of per-voxel (non-constant) levels below it
__device__ __forceinline__ uint32_t lds_level(uint32_t cmask, uint32_t level) { return __popc(~cmask & ((1u << level) - 1u)); }

// The block-constant stack levels' values: lane l of one VGPR holds level l's (workgroup-uniform; every wave keeps its own copy)
__device__ __forceinline__ float cv_get(float cv, uint32_t level) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cv), (int)level)); }
__device__ __forceinline__ void cv_set(float& cv, uint32_t level, float x) { cv = (threadIdx.x & 63u) == level ? x : cv; }

// ---- a leaf combined straight into the level below it ------------------------------------------------------------------------------
// Most combinations take a bare leaf as their second operand (three out of four on the 512^3 asteroid, all on the plates): written to an LDS
// level of its own only to be read back by the next step. Evaluated into registers and combined on the spot, eight voxels of the column at a
// time, the leaf needs no level — many programs then fit ONE 16 KB level, which is what decides how many workgroups a CU holds
// (k_sdf_eval<2>) — and the program runs one step less: the evaluator's scalar bookkeeping per step weighs as much as its vector arithmetic
// (SQ_INSTS_SALU: one scalar instruction per two vector ones, and a CU issues one scalar instruction per cycle). Same arithmetic per voxel,
// in the same order; only combinations that are applied unconditionally (OP_COMBINE) are fused — one behind the 14-position test needs
// its operand's values before it knows whether to apply. (Keeping the leaf's sixteen values in registers ACROSS steps, so that combinations
// with constants could be folded in as well, was tried: every program fitted one level then, but the copies and flags it took cost more
// vector and scalar instructions than the levels it saved.)
struct LeafRun {  // a leaf's evaluation along a thread's column, resumable
    uint32_t kind;
    bool zonly;
    V3 pos, dz;
    float c0, c1, c2;  // zonly: [kind 0/1] x^2 + y'^2, radius; [kind 2] px^2 + py^2, max(qx, qy), half extent z — else the node's a, b, c
    bool inside;       // zonly box: the wave's columns all lie inside the box's footprint (box_column_inside)
};
__device__ __forceinline__ LeafRun leaf_begin(const ivx_sdf_processed_node* nd, uint32_t kind, V3 origin_root, uint32_t ti, uint32_t tj) {
    const float* m = nd->transform;
    const V3 origin = xform_point(m, origin_root);
    const V3 dx = mk(m[0], m[1], m[2]), dy = mk(m[4], m[5], m[6]);
    LeafRun r;
    r.kind = kind;
    r.dz = mk(m[8], m[9], m[10]);
    const V3 opx = add(origin, scale(dx, (float)ti));
    r.pos = add(opx, scale(dy, (float)tj));
    r.zonly = r.dz.x == 0.0f && r.dz.y == 0.0f;
    const float pa = nd->a, pb = nd->b, pc = nd->c;
    r.c0 = pa, r.c1 = pb, r.c2 = pc;
    r.inside = false;
    if (r.zonly) {  // (eval_leaf's column constants)
        if (kind == 2u) {
            const float qx = fabsf(r.pos.x) - pa, qy = fabsf(r.pos.y) - pb;
            const float px = max_rs(qx, 0.0f), py = max_rs(qy, 0.0f);
            r.c0 = px * px + py * py, r.c1 = max_rs(qx, qy), r.c2 = pc;
            r.inside = box_column_inside(r.c0, r.pos.z, r.dz.z, pc);
        } else {
            float y = r.pos.y, rad = pa;
            if (kind == 1u) {
                float c = y;
                if (c < -pa) c = -pa;
                if (c > pa) c = pa;
                y -= c;
                rad = pb;
            }
            r.c0 = r.pos.x * r.pos.x + y * y, r.c1 = rad;
        }
    }
    return r;
}
// the next eight voxels of the column
__device__ __forceinline__ void leaf_next8(LeafRun& r, float* v) {
    if (r.zonly) {
        float z = r.pos.z;
        if (r.kind == 2u && r.inside) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float qz = fabsf(z) - r.c2, pz = max_rs(qz, 0.0f);
                v[k] = pz + min_rs(max_rs(r.c1, qz), 0.0f);
                z += r.dz.z;
            }
        } else if (r.kind == 2u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float qz = fabsf(z) - r.c2, pz = max_rs(qz, 0.0f);
                v[k] = sqrt_rn(r.c0 + pz * pz) + min_rs(max_rs(r.c1, qz), 0.0f);
                z += r.dz.z;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = sqrt_rn(r.c0 + z * z) - r.c1;
                z += r.dz.z;
            }
        }
        r.pos.z = z;
        return;
    }
    V3 pos = r.pos;
    if (r.kind == 0u) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = len3(pos) - r.c0;
            pos = add(pos, r.dz);
        }
    } else if (r.kind == 1u) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            V3 q = pos;
            float c = q.y;
            if (c < -r.c0) c = -r.c0;
            if (c > r.c0) c = r.c0;
            q.y -= c;
            v[k] = len3(q) - r.c1;
            pos = add(pos, r.dz);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            V3 q = mk(fabsf(pos.x) - r.c0, fabsf(pos.y) - r.c1, fabsf(pos.z) - r.c2);
            V3 qp = mk(max_rs(q.x, 0.0f), max_rs(q.y, 0.0f), max_rs(q.z, 0.0f));
            v[k] = len3(qp) + min_rs(max_rs(max_rs(q.x, q.y), q.z), 0.0f);
            pos = add(pos, r.dz);
        }
    }
    r.pos = pos;
}
// eight voxels (rows 8 h .. 8 h + 7 of the column at d) of `first operand (combination) leaf`: the first operand the constant v1 (`c1`) or the
// level at d, the result to d. `t15`: the level keeps its row 15 in the register r15 (IVX_LV_GET).
template <int KIND, bool SMOOTH>
__device__ __forceinline__ void fused_rows8(float* d, int h, bool t15, float& r15, bool c1, float v1, const float* v, float s, float q) {
    float* dd = d + h * (8 * 256);
    if (c1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float res = combine_t<KIND, SMOOTH>(v1, v[k], s, q);
            if (k == 7 && h == 1 && t15) r15 = res;
            else dd[k * 256] = res;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float a = (k == 7 && h == 1 && t15) ? r15 : dd[k * 256];
            const float res = combine_t<KIND, SMOOTH>(a, v[k], s, q);
            if (k == 7 && h == 1 && t15) r15 = res;
            else dd[k * 256] = res;
        }
    }
}
__device__ __forceinline__ void eval_leaf_fused(const ivx_sdf_processed_node* nd, uint32_t leaf_kind, uint32_t kind, float s, float q, bool c1, float v1, float* d,
                                                bool t15, float& r15, V3 origin_root, uint32_t ti, uint32_t tj) {
    LeafRun run = leaf_begin(nd, leaf_kind, origin_root, ti, tj);
    const bool smooth = s != 0.0f;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        float v[8];
        leaf_next8(run, v);
        if (kind == 7u) {
            if (smooth) fused_rows8<7, true>(d, h, t15, r15, c1, v1, v, s, q);
            else fused_rows8<7, false>(d, h, t15, r15, c1, v1, v, s, q);
        } else if (kind == 8u) {
            if (smooth) fused_rows8<8, true>(d, h, t15, r15, c1, v1, v, s, q);
            else fused_rows8<8, false>(d, h, t15, r15, c1, v1, v, s, q);
        } else {
            if (smooth) fused_rows8<9, true>(d, h, t15, r15, c1, v1, v, s, q);
            else fused_rows8<9, false>(d, h, t15, r15, c1, v1, v, s, q);
        }
    }
}

// Combination of levels top-1 and top (after the caller decremented `top`). `outside` = the node's
// domain lies outside the block, so the reference applies it only if one of the 14 distinct block test
// positions fails `value >= margin` (atomic.rs:788-806, 1661-1797).
template <bool TRIM>
__device__ __forceinline__ void combine_levels(uint32_t kind, float s, float q, float margin, bool outside, uint32_t top, float* stack,
                                               float& cv, uint32_t& cmask, uint32_t tid, float& r15, float* s_pub) {
    // LDS levels are dense: a block-constant stack level is a scalar and takes none (lds_level)
    const bool c1 = (cmask >> (top - 1)) & 1u, c2 = (cmask >> top) & 1u;
    const uint32_t la = lds_level(cmask, top - 1), lb = lds_level(cmask, top);
    float* d1 = stack + (size_t)la * IVX_CHUNK_VOXELS;
    const float* d2 = stack + (size_t)lb * IVX_CHUNK_VOXELS;
    // row 15 in registers (IVX_LV_GET): t1 for dense level la — the first operand's, and the RESULT's also when that operand is a constant
    // (la == lb then: the result takes the second operand's place) —, t2 for the second operand's
    const bool t1 = TRIM && la == 1u, t2 = TRIM && !c2 && lb == 1u;
    const float v1 = c1 ? cv_get(cv, top - 1) : 0.0f, v2 = c2 ? cv_get(cv, top) : 0.0f;
    bool apply = !outside;
    if (!apply) {  // workgroup-uniform branch
        if (c1 && c2) {
            apply = __builtin_amdgcn_readfirstlane(!(combine(kind, v1, v2, s, q) >= margin) ? 1 : 0) != 0;
        } else {
            if (TRIM && (t1 || t2)) {  // the five test voxels of row 15 live in their owners' registers: published for the test lanes
                const uint32_t slot = tid == 0u ? 0u : (tid == 240u ? 1u : (tid == 15u ? 2u : (tid == 255u ? 3u : (tid == 136u ? 4u : 5u))));
                if (slot < 5u) s_pub[slot] = r15;
            }
            __syncthreads();
            // The 26 test positions fall on 14 distinct voxels: 8 corners + 6 face centres. One LANE per voxel (every wave repeats it:
            // the decision has to be known to all of them) instead of every thread walking all 14 — the conjunction does not depend on
            // the order. Lanes 0..7: corner with bit b of the lane choosing 0 or 15 along axis b; lanes 8..13: axis (l - 8) / 2 at 0 or
            // 15, the other two at 8.
            const uint32_t l = tid & 63u;
            uint32_t pi, pj, pk;
            if (l < 8u) {
                pi = (l & 1u) ? 15u : 0u, pj = (l & 2u) ? 15u : 0u, pk = (l & 4u) ? 15u : 0u;
            } else {
                const uint32_t axis = (l - 8u) >> 1, side = ((l - 8u) & 1u) ? 15u : 0u;
                pi = axis == 0u ? side : 8u, pj = axis == 1u ? side : 8u, pk = axis == 2u ? side : 8u;
            }
            bool pass = true;
            if (l < 14u) {
                const uint32_t off = pk * 256u + (pi * 16u + pj);
                // (row 15 of a register-held level: corners 4..7 -> slots 0..3 in the order of the owners above, the z-up face centre -> 4)
                const bool pub = TRIM && pk == 15u;
                const uint32_t ps = l < 8u ? l - 4u : 4u;
                const float x1 = c1 ? v1 : ((pub && t1) ? s_pub[ps] : d1[off]);
                const float x2 = c2 ? v2 : ((pub && t2) ? s_pub[ps] : d2[off]);
                pass = combine(kind, x1, x2, s, q) >= margin;
            }
            const bool all_pass = __all(pass ? 1 : 0) != 0;
            apply = __builtin_amdgcn_readfirstlane(all_pass ? 0 : 1) != 0;
            __syncthreads();
        }
    }
    if (apply) {
        if (c1 && c2) {
            cv_set(cv, top - 1, combine(kind, v1, v2, s, q));
        } else {
            apply_rows_dispatch(kind, d1 + tid, t1, d2 + tid, t2, r15, c1, c2, v1, v2, s, q);
            cmask &= ~(1u << (top - 1));
        }
    }
}

// MODE 0: the general class (as many levels as the program's stack). MODE 1 (TRIM): the two-level class, with row 15 of the second dense level in registers (IVX_LV_GET); `scratch_off`: offset (floats) of sixteen
// words of LDS behind / at the tail of the stack: [0..5) the published test voxels of a register row, [8..12) the classification's votes
// MODE 2: the one-level class (programs whose only operands with a level of their own are fused away, see eval_leaf_fused): 16 KB + the
// scratch words, eight workgroups per CU.
// (eight waves per SIMD: the compiler keeps the scalar registers under the 96 that allows — the kernel's 106 held it to seven, i.e. seven of the
// eight workgroups a CU's LDS takes in the one-level class)
#ifndef IVX_EVAL_WAVES
#define IVX_EVAL_WAVES 8
#endif
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IVX_EVAL_WAVES, 8))) void k_sdf_eval(SampleParams p, const uint32_t* __restrict__ eval_count, const uint32_t* __restrict__ eval_list,
                                                  const uint32_t* __restrict__ long_count, const uint32_t* __restrict__ first_count,
                                                  const uint32_t* __restrict__ first_list, uint32_t list_len, uint32_t scratch_off,
                                                  const uint32_t* __restrict__ prog_len, const uint2* __restrict__ prog_ops,
                                                  const ivx_sdf_processed_node* __restrict__ nodes, int8_t* __restrict__ sdf_out,
                                                  uint8_t* __restrict__ type_out, ivx_chunk_info* __restrict__ info_out,
                                                  uint16_t* __restrict__ signs_out, uint8_t* __restrict__ kface_out,
                                                  const ivx_chunk_info* __restrict__ shadow, uint32_t n_chunks, ivx_roles::PresetArgs preset) {
    extern __shared__ float stack[];  // [stack_size][16][256]
    constexpr bool TRIM = MODE == 1;
    const uint32_t tid = threadIdx.x;
    const uint32_t ti = tid >> 4, tj = tid & 15u;
    if (shadow) {
        // The sample stage's pre-pass ran a step ahead (ivx_grid_set_sample_ahead) and this is the stage's first launch: it hosts what the pre-pass
        // hosts otherwise — the presets of the later stages' scratch words — and commits the records of the chunks the pre-pass settled (Void /
        // Uniform: they were parked in the shadow array while the step before still read the real one), a slice of the grid per workgroup.
        for (uint32_t gid = blockIdx.x * 256u + tid; gid < max(preset.n_sn, 32u); gid += gridDim.x * 256u) ivx_roles::role_preset(preset, gid);  // (a short list's launch has few threads)
        const uint32_t each = (n_chunks + gridDim.x - 1u) / gridDim.x;
        const uint32_t c0 = blockIdx.x * each, c1 = min(c0 + each, n_chunks);
        for (uint32_t c = c0 + tid; c < c1; c += 256u) {
            const ivx_chunk_info r = shadow[c];
            if (r.kind != AHEAD_OPEN) info_out[c] = r;
        }
    }
    // (`first_list`: a second list walked ahead of the launch's own — the host merges two classes into one launch when one of them is too
    // short to fill the chip on its own, see ivx_launch_sdf_sample)
    const uint32_t n_first = first_count ? first_count[0] : 0u;
    const uint32_t n_eval = n_first + eval_count[0];
    const uint32_t n_long = long_count ? long_count[0] : eval_count[0];  // (a list with a long / short split keeps its short entries at the back)
    // bounded grid-stride walk over the list of chunks to evaluate
    for (uint32_t li = blockIdx.x; li < n_eval; li += gridDim.x) {
    __syncthreads();  // the previous chunk's LDS use is over
    IVX_TE(p, li, 0);
    const uint32_t lj = li - n_first;
    const uint32_t chunk = li < n_first ? first_list[li] : eval_list[lj < n_long ? lj : list_len - 1u - (lj - n_long)];
    const uint32_t ck = chunk % p.cz, cj = (chunk / p.cz) % p.cy, ci = chunk / (p.cz * p.cy);
    const uint32_t oi = (ci + p.x_off) * 16u, oj = cj * 16u, ok = ck * 16u;
    int sd[16];

    const V3 origin_root = sub(mk((float)oi, (float)oj, (float)ok), mk(p.shifted_center[0], p.shifted_center[1], p.shifted_center[2]));
    const Box block{origin_root, add(origin_root, mk(16.0f, 16.0f, 16.0f))};

    // Control flow is wave-uniform and scalar. Block-constant propagation: a stack level whose 4096
    // values are all equal (`fill(+-margin)`) is kept as one scalar; combining two constants yields a
    // constant. Element-wise results are unchanged (same op on equal inputs).
    float cv = 0.0f;  // values of the block-constant levels (cv_get / cv_set)
    float r15 = 0.0f;  // TRIM: row 15 of the second dense level (IVX_LV_GET)
    // The compact program's last step, when it scales the per-voxel root level or combines it with a constant, is not run as a step of its
    // own — a read and a write of the level for one multiplication or comparison per voxel — but applied as the root level is read for
    // quantisation (same operations on the same values): 5 = scaling, 7 / 8 / 9 = applied combination with the constant `tail_c` as
    // second operand. (Nearly every program ends this way: a body under a root Scaling; a far body or far holes folded to one constant.)
    uint32_t tail = 0u;
    float tail_c = 0.0f, tail_s = 0.0f, tail_q = 0.0f;
    float* s_pub = stack + scratch_off;
    const uint32_t lane = tid & 63u;
    uint32_t top = 0;
    uint32_t cmask = 0;  // bit l set: level l is block-constant, value in lane l of cv

    const uint32_t len = __builtin_amdgcn_readfirstlane(prog_len[chunk]);
    if (len != OP_OVERFLOW) {
        // ---- compact program written by the pre-pass for this chunk: only the steps that need per-voxel
        // work plus folded constants. Lane i holds op i (two rounds for OP_CAP = 128) and the scalar
        // parameters of its node, so the serial walk below never waits on memory.
        const uint2* ops = prog_ops + (size_t)chunk * OP_CAP;
        uint32_t my_w[2], my_v[2];
        float my_s[2], my_q[2], my_m[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t i = lane + 64u * r;
            uint2 o = make_uint2(0u, 0u);
            if (i < len) o = ops[i];
            my_w[r] = o.x;
            my_v[r] = o.y;
            const uint32_t opc = o.x >> 28;
            const ivx_sdf_processed_node* nd = nodes + (o.x & 0xFFFFFFu);
            my_s[r] = 0.0f, my_q[r] = 0.0f, my_m[r] = 0.0f;
            if (i < len && opc >= OP_SCALE) {
                my_s[r] = nd->a;
                my_q[r] = nd->b;
                my_m[r] = nd->margin;
            }
        }
        IVX_TE(p, li, 1);  // program fetched
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t l = i & 63u;
            uint32_t w, v;
            float s, q, margin;
            if (i < 64u) {
                w = __builtin_amdgcn_readlane(my_w[0], l);
                v = __builtin_amdgcn_readlane(my_v[0], l);
                s = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_s[0]), l));
                q = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_q[0]), l));
                margin = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_m[0]), l));
            } else {
                w = __builtin_amdgcn_readlane(my_w[1], l);
                v = __builtin_amdgcn_readlane(my_v[1], l);
                s = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_s[1]), l));
                q = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_q[1]), l));
                margin = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_m[1]), l));
            }
            const uint32_t opc = w >> 28, kind = (w >> 24) & 15u;
            if (opc == OP_CONST) {
                if (i + 2u == len && top == 1u && !(cmask & 1u)) {  // (the last step but one, over a per-voxel root level: a tail combination?)
                    const uint32_t l1 = (i + 1u) & 63u;
                    const uint32_t wn = i + 1u < 64u ? __builtin_amdgcn_readlane(my_w[0], l1) : __builtin_amdgcn_readlane(my_w[1], l1);
                    if ((wn >> 28) == OP_COMBINE) {
                        tail = (wn >> 24) & 15u;
                        tail_c = __uint_as_float(v);
                        tail_s = __uint_as_float(i + 1u < 64u ? __builtin_amdgcn_readlane(__float_as_uint(my_s[0]), l1) : __builtin_amdgcn_readlane(__float_as_uint(my_s[1]), l1));
                        tail_q = __uint_as_float(i + 1u < 64u ? __builtin_amdgcn_readlane(__float_as_uint(my_q[0]), l1) : __builtin_amdgcn_readlane(__float_as_uint(my_q[1]), l1));
                        break;
                    }
                }
                cv_set(cv, top, __uint_as_float(v));
                cmask |= 1u << top;
                top += 1;
            } else if (opc == OP_LEAF) {
                // a leaf that the next step combines unconditionally with the level below is combined from registers (eval_leaf_fused; the
                // pre-pass counted the chunk's levels on the same rule)
                uint32_t wn = 0u;
                float sn = 0.0f, qn = 0.0f;
                if (i + 1u < len) {
                    const uint32_t l1 = (i + 1u) & 63u;
                    if (i + 1u < 64u) {
                        wn = __builtin_amdgcn_readlane(my_w[0], l1);
                        sn = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_s[0]), l1));
                        qn = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_q[0]), l1));
                    } else {
                        wn = __builtin_amdgcn_readlane(my_w[1], l1);
                        sn = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_s[1]), l1));
                        qn = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_q[1]), l1));
                    }
                }
                if ((wn >> 28) == OP_COMBINE && top >= 1u) {
                    const bool c1 = (cmask >> (top - 1)) & 1u;
                    const float v1 = c1 ? cv_get(cv, top - 1) : 0.0f;
                    cmask &= ~(1u << (top - 1));  // the result is per voxel, in the first operand's place
                    const uint32_t dl = lds_level(cmask, top - 1);
                    eval_leaf_fused(nodes + (w & 0xFFFFFFu), kind, (wn >> 24) & 15u, sn, qn, c1, v1, stack + (size_t)dl * IVX_CHUNK_VOXELS + tid, TRIM && dl == 1u, r15,
                                    origin_root, ti, tj);
                    i += 1u;  // (the combination is done)
                } else {
                    cmask &= ~(1u << top);
                    {
                        const uint32_t dl = lds_level(cmask, top);
                        eval_leaf(nodes + (w & 0xFFFFFFu), kind, stack + (size_t)dl * IVX_CHUNK_VOXELS + tid, TRIM && dl == 1u, r15, origin_root, ti, tj);
                    }
                    top += 1;
                }
            } else if (opc == OP_SCALE) {
                // Levels the pre-pass KNEW to be constant were folded there. A level can still be a constant here: an
                // OP_COMBINE_OUTSIDE over (constant, per-voxel) that the 14 test positions decide not to apply leaves its
                // constant first operand standing (combine_levels), which the pre-pass cannot foresee.
                if ((cmask >> (top - 1)) & 1u) {
                    cv_set(cv, top - 1, cv_get(cv, top - 1) * s);
                } else if (i + 1u == len && top == 1u) {
                    tail = 5u;  // (the last step, over the per-voxel root level)
                    tail_s = s;
                } else {
                    const uint32_t dl = lds_level(cmask, top - 1);
                    float* d = stack + (size_t)dl * IVX_CHUNK_VOXELS + tid;
                    const bool t15 = TRIM && dl == 1u;
#pragma unroll
                    for (int k = 0; k < 16; ++k) IVX_LV_SET(d, k, t15, r15, IVX_LV_GET(d, k, t15, r15) * s);
                }
            } else if (opc == OP_SKIP) {
                i += v - 1u;  // (a first operand the pre-pass found out of its combination's reach)
            } else {
                top -= 1;
                combine_levels<TRIM>(kind, s, q, margin, opc == OP_COMBINE_OUTSIDE, top, stack, cv, cmask, tid, r15, s_pub);
            }
        }
    } else {
        // ---- fallback: the full node program; every node's block-level decision (the reference's
        // whole-block early-outs) is taken by one LANE per node, 64 nodes at a time, and broadcast as
        // two 64-bit ballot masks held in SGPRs.
        for (uint32_t g0 = 0; g0 < p.n_nodes; g0 += 64u) {
            const uint32_t mine = g0 + lane;
            const uint32_t my_mode = mine < p.n_nodes ? node_mode(nodes + mine, block) : 0u;
            const unsigned long long mk1 = __ballot(my_mode == 1u), mk2 = __ballot(my_mode == 2u);
            const uint32_t cnt = min(64u, p.n_nodes - g0);
            for (uint32_t j = 0; j < cnt; ++j) {
                const ivx_sdf_processed_node* nd = nodes + g0 + j;
                const uint32_t kind = __builtin_amdgcn_readfirstlane(nd->kind);
                const uint32_t mode = ((mk1 >> j) & 1ull) ? 1u : (((mk2 >> j) & 1ull) ? 2u : 0u);
                if (kind <= 2u) {
                    if (mode != 0u) {
                        cv_set(cv, top, mode == 1u ? nd->margin : -nd->margin);
                        cmask |= 1u << top;
                    } else {
                        cmask &= ~(1u << top);
                        {
                            const uint32_t dl = lds_level(cmask, top);
                            eval_leaf(nd, kind, stack + (size_t)dl * IVX_CHUNK_VOXELS + tid, TRIM && dl == 1u, r15, origin_root, ti, tj);
                        }
                    }
                    top += 1;
                } else if (kind == 5u) {
                    const float s = nd->a;
                    if ((cmask >> (top - 1)) & 1u) {
                        cv_set(cv, top - 1, cv_get(cv, top - 1) * s);
                    } else {
                        const uint32_t dl = lds_level(cmask, top - 1);
                        float* d = stack + (size_t)dl * IVX_CHUNK_VOXELS + tid;
                        const bool t15 = TRIM && dl == 1u;
#pragma unroll
                        for (int k = 0; k < 16; ++k) IVX_LV_SET(d, k, t15, r15, IVX_LV_GET(d, k, t15, r15) * s);
                    }
                } else if (kind >= 7u) {
                    top -= 1;
                    combine_levels<TRIM>(kind, nd->a, nd->b, nd->margin, mode != 0u, top, stack, cv, cmask, tid, r15, s_pub);
                }
            }
        }
    }

    IVX_TE(p, li, 2);  // program evaluated
    IVX_TE(p, li, 3);
    const bool root_const = cmask & 1u;
    const float root_val = root_const ? cv_get(cv, 0u) : 0.0f;
    {
        float fv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) fv[k] = root_const ? root_val : stack[k * 256 + tid];
        if (tail == 5u) {  // (workgroup-uniform branches: one loop runs)
#pragma unroll
            for (int k = 0; k < 16; ++k) fv[k] = fv[k] * tail_s;
        } else if (tail != 0u) {
            if (tail_s != 0.0f) {
#pragma unroll
                for (int k = 0; k < 16; ++k) fv[k] = combine(tail, fv[k], tail_c, tail_s, tail_q);
            } else if (tail == 7u) {
#pragma unroll
                for (int k = 0; k < 16; ++k) fv[k] = min_rs(fv[k], tail_c);
            } else if (tail == 8u) {
#pragma unroll
                for (int k = 0; k < 16; ++k) fv[k] = max_rs(fv[k], -tail_c);
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) fv[k] = max_rs(fv[k], tail_c);
            }
        }
        if (oi + 16u <= p.shape[0] && oj + 16u <= p.shape[1] && ok + 16u <= p.shape[2]) {  // (workgroup-uniform) the chunk lies inside the grid
#pragma unroll
            for (int k = 0; k < 16; ++k) sd[k] = sd_from_f32(fv[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                bool in_grid = (oi + ti) < p.shape[0] && (oj + tj) < p.shape[1] && (ok + (uint32_t)k) < p.shape[2];
                sd[k] = in_grid ? sd_from_f32(fv[k]) : 127;
            }
        }
    }
    IVX_TE(p, li, 4);
    classify_and_store(sd, make_uint4(0, 0, 0, 0), true, 0, sdf_out, type_out, info_out, chunk, tid, true, p.voxel_type, true,
                       reinterpret_cast<uint32_t*>(s_pub) + 8, signs_out, kface_out);
    IVX_TE(p, li, 5);  // classified and stored
    }
}

// Classification of uploaded dense voxels (ChunkedVoxelGenerator contract, generation.rs:41-67).
__global__ __launch_bounds__(256) void k_classify(uint32_t n_chunks, int8_t* __restrict__ sdf, uint8_t* __restrict__ type,
                                                  ivx_chunk_info* __restrict__ info) {
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = ivx_xcd_remap(blockIdx.x, n_chunks);
    size_t base = (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)tid * 16;
    uint4 s = *reinterpret_cast<const uint4*>(sdf + base);
    uint4 t = *reinterpret_cast<const uint4*>(type + base);
    const uint8_t first_type = type[(size_t)chunk * IVX_CHUNK_VOXELS];
    uint32_t sw[4] = {s.x, s.y, s.z, s.w};
    int sd[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) sd[k] = (int)(int8_t)((sw[k >> 2] >> (8 * (k & 3))) & 0xFF);
    uint32_t ft = first_type * 0x01010101u;
    bool types_uniform = t.x == ft && t.y == ft && t.z == ft && t.w == ft;
    __shared__ uint32_t s_votes[4];
    classify_and_store(sd, t, types_uniform, first_type, sdf, type, info, chunk, tid, false, 0, false, s_votes);
}

}  // namespace

int ivx_sampler_buffers(ivx_grid* g) {
    if (g->samp_ops) return IVX_OK;
    // [n] program lengths, [16] counters of the three evaluation lists (+ the long / short split of the first; [8..13) their rolled copy), [3 n] the lists
    IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->samp_len), sizeof(uint32_t) * (4 * (size_t)g->n_chunks + 16)));
    IVX_HIP_CHECK(ivx_memset_async(g->samp_len + g->n_chunks, 0, 16 * sizeof(uint32_t), g->ctx->stream));  // the counters start at zero
    IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->samp_ops), sizeof(uint2) * (size_t)OP_CAP * g->n_chunks));
    return IVX_OK;
}

// the context's second stream (sample-ahead), made on first use
static int ivx_aux_stream(ivx_ctx* c, hipStream_t* out) {
    // (a stream of the default priority: with the lowest priority the pre-pass was served late whenever the process held other queues — under
    // torch.distributed + RCCL the sample stage then waited for it, 0.247 ms per slab step against 0.200 — and gained nothing elsewhere)
    if (!c->aux_stream) IVX_HIP_CHECK(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    *out = c->aux_stream;
    return IVX_OK;
}
// second set of the sampler's buffers + the shadow records + the two events of the hand-over
static int ivx_sampler_ahead_buffers(ivx_grid* g, hipStream_t aux) {
    if (!g->ahead_events_ready) {
        IVX_HIP_CHECK(hipEventCreateWithFlags(&g->ahead_go, hipEventDisableTiming));
        IVX_HIP_CHECK(hipEventCreateWithFlags(&g->ahead_done, hipEventDisableTiming));
        g->ahead_events_ready = 1;
    }
    if (g->samp_ops_alt) return IVX_OK;
    IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->samp_len_alt), sizeof(uint32_t) * (4 * (size_t)g->n_chunks + 16)));
    IVX_HIP_CHECK(hipMemsetAsync(g->samp_len_alt + g->n_chunks, 0, 16 * sizeof(uint32_t), aux));
    g->alt_eval_dirty = 0;
    IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->samp_ops_alt), sizeof(uint2) * (size_t)OP_CAP * g->n_chunks));
    IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->info_shadow), 2 * sizeof(ivx_chunk_info) * (size_t)g->n_chunks));  // (two: one being committed, one being written)
    return IVX_OK;
}
// A pre-pass that runs ahead is waited for and forgotten (the resident program changes, the grid goes): what it wrote into the second set is dropped
int ivx_sampler_ahead_cancel(ivx_grid* g) {
    if (!g->ahead_pending) return IVX_OK;
    g->ahead_pending = 0;
    g->alt_eval_dirty = 1;
    IVX_HIP_CHECK(hipEventSynchronize(g->ahead_done));
    return IVX_OK;
}
void ivx_sampler_ahead_free(ivx_grid* g) {
    (void)ivx_sampler_ahead_cancel(g);
    if (g->samp_len_alt) (void)hipFree(g->samp_len_alt);
    if (g->samp_ops_alt) (void)hipFree(g->samp_ops_alt);
    if (g->info_shadow) (void)hipFree(g->info_shadow);
    g->samp_len_alt = nullptr, g->samp_ops_alt = nullptr, g->info_shadow = nullptr;
    if (g->ahead_events_ready) {
        (void)hipEventDestroy(g->ahead_go);
        (void)hipEventDestroy(g->ahead_done);
        g->ahead_events_ready = 0;
    }
}

ivx_roles::PresetArgs ivx_preset_args(ivx_grid* g, uint32_t groups);  // derive.hip

int ivx_launch_sdf_sample(ivx_grid* g, const ivx_sdf_processed_node* d_nodes, uint32_t n_nodes, uint32_t stack_size,
                          const uint32_t shape[3], const float shifted_center[3], uint8_t voxel_type, uint32_t preset_groups,
                          bool resident_program) {
    SampleParams p;
    p.cx = g->cc[0];
    p.cy = g->cc[1];
    p.cz = g->cc[2];
    p.x_off = g->x_off;
    for (int d = 0; d < 3; ++d) {
        p.shape[d] = shape[d];
        p.shifted_center[d] = shifted_center[d];
    }
    p.n_nodes = n_nodes;
    p.stack_size = stack_size;
#ifdef IVX_WG_TRACE
    p.trace = reinterpret_cast<unsigned long long*>(g->chunk_moments);
#endif
    p.voxel_type = voxel_type;
    size_t lds = (size_t)(stack_size ? stack_size : 1) * IVX_CHUNK_VOXELS * sizeof(float);
    IVX_REQUIRE(lds <= 150 * 1024, IVX_ERR_CAPACITY, "SDF graph needs a forward stack of %u blocks (at most 9 fit the 160 KiB LDS)", stack_size);
    IVX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sdf_eval<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        int rc_b = ivx_sampler_buffers(g);
        if (rc_b) return rc_b;
    }
    // Programs of up to SUPER_MAX_WORDS * 32 nodes get their super-block tables inside the pre-pass (no launch of their own); the
    // pre-pass is then the call's first kernel and hosts the presets of the later stages' scratch words. The sampler's own list
    // counters are zero already when the derive sweep of the step before rolled them over (role_preset), else cleared here.
    const uint32_t words = (n_nodes + 31u) / 32u > 0u ? (n_nodes + 31u) / 32u : 1u;
    const bool fused_super = words <= SUPER_MAX_WORDS;
    const uint32_t sx = (g->cc[0] + SUPER - 1) / SUPER, sy = (g->cc[1] + SUPER - 1) / SUPER, sz = (g->cc[2] + SUPER - 1) / SUPER;
    // One evaluator launch per LDS class (see k_sdf_prepass); a class the program cannot reach is not launched, nor one whose list the last step
    // under this program found empty (the lists are a function of the program and the grid). Two launches one after the other each pay
    // their own ramp and tail: when both of the first two classes have work and one of them is too short to fill the chip a few times
    // over, the two-level kernel takes both lists in one launch (the one-level programs run there as well, five workgroups per CU).
    // (`resident_program`: the step path. The remembered lengths are the RESIDENT program's; ivx_sdf_sample runs any program through
    // this launcher and must launch every class that program can reach)
    const bool known = resident_program && g->eval_len_valid != 0;
    const uint32_t fill = 4u * 5u * (uint32_t)g->ctx->n_cu;
    const bool merge01 = known && g->eval_len[0] && g->eval_len[1] && (g->eval_len[0] < fill || g->eval_len[1] < fill);
    const bool launch2 = !merge01 && !(known && g->eval_len[0] == 0u);            // k_sdf_eval<2>: the one-level class
    const bool launch1 = merge01 || !(known && g->eval_len[1] == 0u);             // k_sdf_eval<1>: the two-level class (or both)
    const bool launch0 = stack_size >= 3u && !(known && g->eval_len[2] == 0u);    // k_sdf_eval<0>: the general class
    // Sample-ahead: this stage's pre-pass may have run already, behind the evaluator of the step before (see below); the stage then starts at
    // its first evaluator launch, which commits the parked records and hosts the presets. Only with an evaluator launch to host them.
    const bool ahead_fits = resident_program && fused_super && (launch0 || launch1 || launch2);
    bool consume = resident_program && g->ahead_pending != 0;
    if (consume && !ahead_fits) {
        int rc_c = ivx_sampler_ahead_cancel(g);
        if (rc_c) return rc_c;
        consume = false;
    }
    const bool eval_dirty = (g->scratch_dirty & IVX_SCRATCH_EVAL) != 0u;
    static const bool ahead_trace = getenv("IVX_AHEAD_TRACE") != nullptr;  // (developer aid: what every sample stage did about its pre-pass)
    if (ahead_trace)
        fprintf(stderr, "[ivx sample] grid %p: resident %d ahead_on %d pending %d fits %d (known %d launches %d%d%d) unordered_ok %d -> %s\n", (void*)g, (int)resident_program,
                g->ahead_on, g->ahead_pending, (int)ahead_fits, (int)known, (int)launch2, (int)launch1, (int)launch0, g->ahead_unordered_ok, consume ? "consume" : "own pre-pass");
    if (consume) {
        std::swap(g->samp_len, g->samp_len_alt);
        void* t_ops = g->samp_ops;
        g->samp_ops = g->samp_ops_alt;
        g->samp_ops_alt = t_ops;
        g->alt_eval_dirty = eval_dirty ? 1 : 0;  // (the set that steps back: its counters are zero if a derive sweep rolled them over)
        g->ahead_pending = 0;
        // (normally long over — it was enqueued a step ago —, and then the stream need not hear of it: a wait is a packet on the queue)
        if (hipEventQuery(g->ahead_done) != hipSuccess) IVX_HIP_CHECK(hipStreamWaitEvent(g->ctx->stream, g->ahead_done, 0));
    }
    uint2* ops = reinterpret_cast<uint2*>(g->samp_ops);
    uint32_t* eval_count = g->samp_len + g->n_chunks;
    uint32_t* eval_list = eval_count + 16;
    const uint32_t host_presets = preset_groups & ~IVX_SCRATCH_EVAL;  // what the stage's first kernel presets for the later stages
    if (!consume) {
        uint32_t super_presets = fused_super ? 0u : preset_groups;
        const uint32_t prepass_presets = fused_super ? host_presets : 0u;
        if (eval_dirty && !(super_presets & IVX_SCRATCH_EVAL)) IVX_HIP_CHECK(ivx_memset_async(eval_count, 0, 8 * sizeof(uint32_t), g->ctx->stream));
        if (!eval_dirty) super_presets &= ~IVX_SCRATCH_EVAL;
        uint2* super_skip = nullptr;
        if (!fused_super) {
            const size_t need = (((size_t)sx * sy * sz * words + 1u) & ~(size_t)1u) + (size_t)sx * sy * sz * words * 64u;  // far bits + a uint2 per node
            if (need > g->samp_super_words) {
                IVX_HIP_CHECK(ivx_stream_sync(g->ctx->stream));
                if (g->samp_super) (void)hipFree(g->samp_super);
                g->samp_super = nullptr;
                g->samp_super_words = 0;
                IVX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->samp_super), need * sizeof(uint32_t)));
                g->samp_super_words = need;
            }
            super_skip = reinterpret_cast<uint2*>(g->samp_super + (((size_t)sx * sy * sz * words + 1u) & ~(size_t)1u));
            IVX_KLAUNCH(k_sdf_super, dim3(sx * sy * sz), dim3(64), words * sizeof(uint32_t), g->ctx->stream, p, d_nodes, g->samp_super, super_skip, words, sy, sz,
                               ivx_preset_args(g, super_presets));
        }
        IVX_KLAUNCH(k_sdf_prepass, dim3(sx * sy * sz), dim3(PRE_T * PRE_WAVES), 0, g->ctx->stream, p, d_nodes, g->samp_len, ops, eval_count, eval_list,
                           g->n_chunks, g->info, g->samp_super, super_skip, words, sy, sz, sx * sy * sz, fused_super ? 1u : 0u, 0u, ivx_preset_args(g, prepass_presets));
    }
    g->scratch_dirty = (g->scratch_dirty & ~preset_groups) | IVX_SCRATCH_EVAL;
    g->planes_compact = 1;
    {
        // (grids: a whole number of list entries per workgroup once the lists' lengths are known — a launch's last wave of workgroups costs as
        // much as a full one, see ivx_launch_derive)
        auto fit = [&](uint32_t n) {
            const uint32_t cap = g->n_chunks < 16384u ? g->n_chunks : 16384u;  // (measured on 512^3: 4 096 and 32 768 are both slower on the all-surface grid)
            if (!known || n == 0u) return cap;
            const uint32_t each = (n + cap - 1u) / cap;
            return (n + each - 1u) / each;
        };
        // (consume: the first launch commits the shadow records and hosts the presets)
        const ivx_chunk_info* shadow = consume ? g->info_shadow + (size_t)g->shadow_sel * g->n_chunks : nullptr;
        const ivx_roles::PresetArgs no_presets = ivx_preset_args(g, 0u), first_presets = ivx_preset_args(g, consume ? host_presets : 0u);
        auto take_shadow = [&]() {
            const ivx_chunk_info* s_ = shadow;
            shadow = nullptr;
            return s_;
        };
        uint32_t* const list0 = eval_list;
        uint32_t* const list1 = eval_list + (size_t)g->n_chunks;
        if (launch2) {
            // one level + 64 words of scratch: 16 640 bytes = 13 LDS granules, eight workgroups per CU (the waves a SIMD holds)
            const uint32_t scratch_off = IVX_CHUNK_VOXELS;
            const ivx_chunk_info* sh = take_shadow();
            IVX_KLAUNCH(k_sdf_eval<2>, dim3(fit(g->eval_len[0])), dim3(256), (size_t)(scratch_off + 64u) * sizeof(float), g->ctx->stream, p, eval_count + 0, list0,
                               eval_count + 3, nullptr, nullptr, g->n_chunks, scratch_off, g->samp_len, ops, d_nodes, g->sdf, g->type, g->info, g->chunk_signs, g->kface,
                               sh, g->n_chunks, sh ? first_presets : no_presets);
        }
        if (launch1) {
            // two levels, the second one 15 rows long + 64 words of scratch: 32 000 bytes = 25 LDS granules, five workgroups per CU
            const uint32_t scratch_off = IVX_CHUNK_VOXELS + 15u * 256u;
            const ivx_chunk_info* sh = take_shadow();
            if (merge01)
                IVX_KLAUNCH(k_sdf_eval<1>, dim3(fit(g->eval_len[0] + g->eval_len[1])), dim3(256), (size_t)(scratch_off + 64u) * sizeof(float), g->ctx->stream, p, eval_count + 0, list0,
                                   eval_count + 3, eval_count + 1, list1, g->n_chunks, scratch_off, g->samp_len, ops, d_nodes, g->sdf, g->type, g->info, g->chunk_signs, g->kface,
                                   sh, g->n_chunks, sh ? first_presets : no_presets);
            else
                IVX_KLAUNCH(k_sdf_eval<1>, dim3(fit(g->eval_len[1])), dim3(256), (size_t)(scratch_off + 64u) * sizeof(float), g->ctx->stream, p, eval_count + 1, list1,
                                   nullptr, nullptr, nullptr, g->n_chunks, scratch_off, g->samp_len, ops, d_nodes, g->sdf, g->type, g->info, g->chunk_signs, g->kface,
                                   sh, g->n_chunks, sh ? first_presets : no_presets);
        }
        if (launch0) {
            // (the scratch words are the last sixteen of the stack: rows 15 of the last level's threads 240..255, dead when they are used
            // — the votes — and never used as published test voxels, which only the trimmed launch has)
            const uint32_t lv = stack_size;
            const uint32_t scratch_off = lv * IVX_CHUNK_VOXELS - 16u;
            const ivx_chunk_info* sh = take_shadow();
            IVX_KLAUNCH(k_sdf_eval<0>, dim3(fit(g->eval_len[2])), dim3(256), (size_t)lv * IVX_CHUNK_VOXELS * sizeof(float), g->ctx->stream, p, eval_count + 2,
                               eval_list + 2 * (size_t)g->n_chunks, nullptr, nullptr, nullptr, g->n_chunks, scratch_off, g->samp_len, ops, d_nodes, g->sdf, g->type, g->info, g->chunk_signs, g->kface,
                               sh, g->n_chunks, sh ? first_presets : no_presets);
        }
    }
    IVX_HIP_CHECK(hipGetLastError());
    g->ahead_wanted = (g->ahead_on && ahead_fits) ? 1 : 0;
    // (everything enqueued before this call has been collected: the sampler's other set of buffers is idle and the pre-pass needs no place in
    // the stream's order; else the step puts it behind its last big launch)
    if (g->ahead_wanted && g->ahead_unordered_ok) {
        const int rc_a = ivx_sampler_launch_ahead(g, false);
        if (rc_a) return rc_a;
    }
    // every chunk that has planes now has its sign rows and k-face bytes too, and one type throughout (SameVoxelTypeGenerator): the derive
    // sweep may work from those (until something else rewrites voxels: ivx_planes_touched)
    g->signs_current = 1;
    g->signs_type = voxel_type;
    return IVX_OK;
}

// The NEXT sample stage's pre-pass, a step ahead (ivx_grid_set_sample_ahead): it reads nothing but the resident program and the grid's
// geometry, so it runs on the context's second stream beside this step's kernels — into the sampler's other set of buffers (this step's
// lists and programs are its evaluator's) and the other shadow record array. At most two blocks per CU (what a CU's LDS holds; developer
// knob IVX_AHEAD_BLOCKS), which walk the super-blocks of a larger grid in turn. Measured on the 512^3 step: 0.217 ms -> 0.199 with a block
// per super-block or per CU alike; a block per four CUs no longer finishes inside the step.
// `behind_stream`: the launch waits for what the context's stream holds now (a caller that has NOT collected its earlier steps: the other
// set may still be read). A step that never comes costs one unused pre-pass; ivx_grid_set_sdf_program and ivx_grid_destroy wait for one
// that is under way.
int ivx_sampler_launch_ahead(ivx_grid* g, bool behind_stream) {
    if (!g->ahead_wanted) return IVX_OK;
    g->ahead_wanted = 0;
    if (g->ahead_pending) return IVX_OK;
    hipStream_t aux;
    int rc = ivx_aux_stream(g->ctx, &aux);
    if (rc) return rc;
    if ((rc = ivx_sampler_ahead_buffers(g, aux))) return rc;
    SampleParams p;
    p.cx = g->cc[0];
    p.cy = g->cc[1];
    p.cz = g->cc[2];
    p.x_off = g->x_off;
    for (int d = 0; d < 3; ++d) {
        p.shape[d] = g->prog_shape[d];
        p.shifted_center[d] = g->prog_center[d];
    }
    p.n_nodes = g->prog_n;
    p.stack_size = g->prog_stack;
#ifdef IVX_WG_TRACE
    p.trace = reinterpret_cast<unsigned long long*>(g->chunk_moments);
#endif
    p.voxel_type = g->prog_type;
    const uint32_t words = (g->prog_n + 31u) / 32u > 0u ? (g->prog_n + 31u) / 32u : 1u;
    const uint32_t sx = (g->cc[0] + SUPER - 1) / SUPER, sy = (g->cc[1] + SUPER - 1) / SUPER, sz = (g->cc[2] + SUPER - 1) / SUPER;
    if (behind_stream) {
        IVX_HIP_CHECK(hipEventRecord(g->ahead_go, g->ctx->stream));
        IVX_HIP_CHECK(hipStreamWaitEvent(aux, g->ahead_go, 0));
    }
    uint32_t* alt_count = g->samp_len_alt + g->n_chunks;
    if (g->alt_eval_dirty) IVX_HIP_CHECK(hipMemsetAsync(alt_count, 0, 8 * sizeof(uint32_t), aux));
    g->alt_eval_dirty = 0;
    static const uint32_t blocks_env = [] {
        const char* e = getenv("IVX_AHEAD_BLOCKS");
        return e ? (uint32_t)strtoul(e, nullptr, 10) : 0u;
    }();
    const uint32_t n_sb = sx * sy * sz, blocks = std::min(n_sb, blocks_env ? blocks_env : 2u * (uint32_t)g->ctx->n_cu);
    g->shadow_sel ^= 1;
    hipLaunchKernelGGL(k_sdf_prepass, dim3(blocks), dim3(PRE_T * PRE_WAVES), 0, aux, p, g->prog_nodes, g->samp_len_alt, reinterpret_cast<uint2*>(g->samp_ops_alt),
                       alt_count, alt_count + 16, g->n_chunks, g->info_shadow + (size_t)g->shadow_sel * g->n_chunks, nullptr, nullptr, words, sy, sz, n_sb, 1u, 1u,
                       ivx_preset_args(g, 0u));
    IVX_HIP_CHECK(hipGetLastError());
    IVX_HIP_CHECK(hipEventRecord(g->ahead_done, aux));
    g->ahead_pending = 1;
    return IVX_OK;
}

int ivx_launch_classify(ivx_grid* g) {
    ivx_planes_touched(g);
    IVX_KLAUNCH(k_classify, dim3(g->n_chunks), dim3(256), 0, g->ctx->stream, g->n_chunks, g->sdf, g->type, g->info);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
