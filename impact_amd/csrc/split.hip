// a11 — splitting a disconnected region off a voxel object.
//
// Reference (engine/crates/impact_voxel/src/object/extraction.rs): extract_disconnected_region (297-596) walks the
// chunk box of the region and, per chunk,
//   * chunk holds no voxel of the region            -> Void chunk in the new object;
//   * uniform chunk of the region                   -> moved; Void in the parent;
//   * non-uniform chunk with no other region in it  -> moved as a whole (empty voxels included); Void in the parent;
//   * non-uniform chunk shared with other regions   -> per voxel: empty voxels are COPIED (their signed distances
//                                                      shape the mesh), region voxels are MOVED (the parent gets
//                                                      Voxel::maximally_outside()), voxels of other regions become
//                                                      maximally_outside in the new object.
// Objects of at most 2x2x2 chunks whose occupied extent fits 14 voxels are repacked into one chunk with a
// one-voxel empty border (1972-2123). Derived state of both objects is recomputed afterwards by the ordinary
// kernels (derive.hip, ccl.hip): it is a pure function of the voxels and the chunk kinds set here.
//
// One workgroup per chunk of the box; a thread owns a 16-voxel k-row (four 16-byte plane loads).
#include "ivx_internal.hpp"

namespace {

struct SplitParams {
    GridView p;               // parent
    uint32_t lo[3];           // chunk box origin in the parent
    uint32_t cc[3];           // chunk counts of the box (= of the child)
    uint32_t target;          // component id to move
};

__global__ __launch_bounds__(256) void k_split_move(SplitParams sp, const uint8_t* __restrict__ p_labels, const uint32_t* __restrict__ rcompid,
                                                    int8_t* __restrict__ p_sdf, uint8_t* __restrict__ p_type, ivx_chunk_info* __restrict__ p_info,
                                                    int8_t* __restrict__ c_sdf, uint8_t* __restrict__ c_type, ivx_chunk_info* __restrict__ c_info) {
    __shared__ uint32_t s_comp[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t cchunk = blockIdx.x;
    const uint32_t ck = cchunk % sp.cc[2], cj = (cchunk / sp.cc[2]) % sp.cc[1], ci = cchunk / (sp.cc[2] * sp.cc[1]);
    const uint32_t pchunk = ((ci + sp.lo[0]) * sp.p.cy + (cj + sp.lo[1])) * sp.p.cz + (ck + sp.lo[2]);
    const ivx_chunk_info pinfo = p_info[pchunk];
    const size_t pb = (size_t)pchunk * IVX_CHUNK_VOXELS + (size_t)tid * 16, cb = (size_t)cchunk * IVX_CHUNK_VOXELS + (size_t)tid * 16;
    const uint4 void_sd = make_uint4(0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu);
    const uint4 void_ty = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    s_comp[tid] = tid < pinfo.region_count ? rcompid[pchunk * 256u + tid] : 0xFFFFFFFFu;
    __syncthreads();
    uint4 l4 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if (pinfo.kind != KIND_VOID) l4 = *reinterpret_cast<const uint4*>(p_labels + pb);
    const uint32_t lw[4] = {l4.x, l4.y, l4.z, l4.w};
    // (several regions may leave one after the other on ONE labelling, ivx_split_off_all: a voxel an earlier move has taken still carries its
    // label but is maximally outside now — it counts for nobody)
    uint4 s4 = void_sd;
    if (pinfo.kind != KIND_VOID) s4 = *reinterpret_cast<const uint4*>(p_sdf + pb);
    const uint32_t sdw[4] = {s4.x, s4.y, s4.z, s4.w};
    uint32_t in_region = 0, other = 0;  // bit k
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t lab = (lw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        if (lab != 255u && ((sdw[k >> 2] >> (8 * (k & 3))) & 0xFFu) != 0x7Fu) {
            if (s_comp[lab] == sp.target) in_region |= 1u << k;
            else other |= 1u << k;
        }
    }
    const int has_region = __syncthreads_or(in_region != 0);
    const int has_other = __syncthreads_or(other != 0);
    ivx_chunk_info cinfo;
    cinfo.kind = cinfo.gen_kind = KIND_VOID;
    cinfo.flags = 0;
    cinfo.uniform_type = 0;
    cinfo.face_dist = 0;
    cinfo.region_count = cinfo.boundary_region_count = 0;
    if (!has_region) {
        if (c_sdf) {
            *reinterpret_cast<uint4*>(c_sdf + cb) = void_sd;
            *reinterpret_cast<uint4*>(c_type + cb) = void_ty;
            if (tid == 0) c_info[cchunk] = cinfo;
        }
        return;
    }
    const uint4 t4 = *reinterpret_cast<const uint4*>(p_type + pb);
    if (!has_other) {
        // whole chunk changes owner
        if (c_sdf) {
            *reinterpret_cast<uint4*>(c_sdf + cb) = s4;
            *reinterpret_cast<uint4*>(c_type + cb) = t4;
        }
        *reinterpret_cast<uint4*>(p_sdf + pb) = void_sd;
        *reinterpret_cast<uint4*>(p_type + pb) = void_ty;
        if (tid == 0) {
            cinfo.kind = cinfo.gen_kind = pinfo.kind;  // Uniform stays Uniform, (demoted) NonUniform stays NonUniform
            cinfo.uniform_type = pinfo.uniform_type;
            if (c_sdf) c_info[cchunk] = cinfo;
            ivx_chunk_info pv = cinfo;
            pv.kind = pv.gen_kind = KIND_VOID;
            pv.uniform_type = 0;
            p_info[pchunk] = pv;
        }
        return;
    }
    // shared chunk: move the region's voxels one by one
    uint32_t sw[4] = {s4.x, s4.y, s4.z, s4.w}, tw[4] = {t4.x, t4.y, t4.z, t4.w};
    uint32_t csw[4], ctw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        csw[q] = sw[q];
        ctw[q] = tw[q];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t m = 0xFFu << (8 * (k & 3));
        if ((in_region >> k) & 1u) {  // parent loses it
            sw[k >> 2] = (sw[k >> 2] & ~m) | (0x7Fu << (8 * (k & 3)));
            tw[k >> 2] |= m;
        } else if ((other >> k) & 1u) {  // child must not get it
            csw[k >> 2] = (csw[k >> 2] & ~m) | (0x7Fu << (8 * (k & 3)));
            ctw[k >> 2] |= m;
        }
    }
    if (c_sdf) {
        *reinterpret_cast<uint4*>(c_sdf + cb) = make_uint4(csw[0], csw[1], csw[2], csw[3]);
        *reinterpret_cast<uint4*>(c_type + cb) = make_uint4(ctw[0], ctw[1], ctw[2], ctw[3]);
    }
    *reinterpret_cast<uint4*>(p_sdf + pb) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
    *reinterpret_cast<uint4*>(p_type + pb) = make_uint4(tw[0], tw[1], tw[2], tw[3]);
    if (tid == 0) {
        cinfo.kind = cinfo.gen_kind = KIND_NONUNIFORM;
        if (c_sdf) c_info[cchunk] = cinfo;
        ivx_chunk_info pv = pinfo;
        pv.kind = pv.gen_kind = KIND_NONUNIFORM;
        pv.flags = 0;
        p_info[pchunk] = pv;
    }
}

// single-chunk repack (extraction.rs:1972-2123): dst voxel (i,j,k) <- src voxel (off + ijk) of the temporary
// <= 2x2x2-chunk object, maximally outside where the window leaves its grid
__global__ __launch_bounds__(256) void k_split_repack(uint32_t scx, uint32_t scy, uint32_t scz, uint32_t o0, uint32_t o1, uint32_t o2,
                                                      const int8_t* __restrict__ s_sdf, const uint8_t* __restrict__ s_type,
                                                      int8_t* __restrict__ d_sdf, uint8_t* __restrict__ d_type, ivx_chunk_info* __restrict__ d_info) {
    const uint32_t tid = threadIdx.x;
    const uint32_t i = tid >> 4, j = tid & 15u;
    for (uint32_t k = 0; k < 16; ++k) {
        const uint32_t si = o0 + i, sj = o1 + j, sk = o2 + k;
        int8_t sd = 127;
        uint8_t ty = TYPE_DUMMY;
        if (si < scx * 16u && sj < scy * 16u && sk < scz * 16u) {
            const size_t o = ((size_t)(((si >> 4) * scy + (sj >> 4)) * scz + (sk >> 4)) << 12) + (((si & 15u) << 8) | ((sj & 15u) << 4) | (sk & 15u));
            sd = s_sdf[o];
            ty = s_type[o];
        }
        d_sdf[tid * 16 + k] = sd;
        d_type[tid * 16 + k] = ty;
    }
    if (tid == 0) {
        ivx_chunk_info ci;
        ci.kind = ci.gen_kind = KIND_NONUNIFORM;
        ci.flags = 0;
        ci.uniform_type = 0;
        ci.face_dist = 0;
        ci.region_count = ci.boundary_region_count = 0;
        d_info[0] = ci;
    }
}

// ---- a12: polyhedron clip -------------------------------------------------------------------------
// extract_polyhedron_with_property_transferrer (extraction.rs:639-1270) / copy_polyhedron_with_property_computer
// (1301-1768). Per chunk of the box: Void in the parent or wholly beyond an outer plane (shifted out by 2.54) ->
// Void; wholly inside every inner plane (shifted in by 2.56) -> moved / copied unchanged; otherwise per voxel
//   d          = quantise(max over the intersecting planes of n.(centre) - displacement)   (row form, 1771-1804)
//   polyhedron = max(sdf, d)            parent (extract only) = max(sdf, complement(d))
// with chunks that end up with no non-empty voxel and only "void" distances (> 100) dropped to Void.
#define IVX_MAX_CLIP_PLANES 64
struct ClipParams {
    GridView p;
    uint32_t lo[3], cc[3];
    uint32_t n_planes;
    int extract;
    float planes[IVX_MAX_CLIP_PLANES][4];
};

__device__ __forceinline__ float plane_sd(const float* pl, float x, float y, float z) { return ((pl[0] * x + pl[1] * y) + pl[2] * z) - pl[3]; }
__device__ __forceinline__ int sd_clamped(float v) {  // VoxelSignedDistance::from_f32_array (lib.rs:207-216)
    float s = v * 50.0f;
    s = s < -128.0f ? -128.0f : (s > 127.0f ? 127.0f : s);
    return (int)s;
}
__device__ __forceinline__ int sd_complement(int e) {  // lib.rs:266-268
    const int a = e == 127 ? 127 : e + 1;
    return a == -128 ? 127 : -a;
}

__device__ __forceinline__ void clip_body(const ClipParams& cp, int8_t* __restrict__ p_sdf, uint8_t* __restrict__ p_type, ivx_chunk_info* __restrict__ p_info,
                                          int8_t* __restrict__ c_sdf, uint8_t* __restrict__ c_type, ivx_chunk_info* __restrict__ c_info, const uint32_t cchunk) {
    const uint32_t tid = threadIdx.x;
    const uint32_t ck = cchunk % cp.cc[2], cj = (cchunk / cp.cc[2]) % cp.cc[1], ci = cchunk / (cp.cc[2] * cp.cc[1]);
    const uint32_t I = ci + cp.lo[0], J = cj + cp.lo[1], K = ck + cp.lo[2];
    const uint32_t pchunk = (I * cp.p.cy + J) * cp.p.cz + K;
    const ivx_chunk_info pinfo = p_info[pchunk];
    const size_t pb = (size_t)pchunk * IVX_CHUNK_VOXELS + (size_t)tid * 16, cb = (size_t)cchunk * IVX_CHUNK_VOXELS + (size_t)tid * 16;
    const uint4 void_sd = make_uint4(0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu);
    const uint4 void_ty = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    ivx_chunk_info vinfo;
    vinfo.kind = vinfo.gen_kind = KIND_VOID;
    vinfo.flags = 0;
    vinfo.uniform_type = 0;
    vinfo.face_dist = 0;
    vinfo.region_count = vinfo.boundary_region_count = 0;
    // chunk vs planes: identical in every thread (uniform control flow)
    const float blo[3] = {(float)(I * 16u), (float)(J * 16u), (float)(K * 16u)};
    const float bhi[3] = {(float)((I + 1u) * 16u), (float)((J + 1u) * 16u), (float)((K + 1u) * 16u)};
    bool outside = pinfo.kind == KIND_VOID;
    unsigned long long isect = 0ull;
    if (!outside) {
        for (uint32_t q = 0; q < cp.n_planes; ++q) {
            const float* pl = cp.planes[q];
            float mn[3], mx[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const bool neg = (__float_as_uint(pl[d]) >> 31) != 0;
                mn[d] = neg ? bhi[d] : blo[d];
                mx[d] = neg ? blo[d] : bhi[d];
            }
            const float outer[4] = {pl[0], pl[1], pl[2], pl[3] + 2.54f}, inner[4] = {pl[0], pl[1], pl[2], pl[3] - 2.56f};
            if (plane_sd(outer, mn[0], mn[1], mn[2]) > 0.0f) outside = true;
            if (!(plane_sd(inner, mx[0], mx[1], mx[2]) < 0.0f)) isect |= 1ull << q;
        }
    }
    if (outside) {
        *reinterpret_cast<uint4*>(c_sdf + cb) = void_sd;
        *reinterpret_cast<uint4*>(c_type + cb) = void_ty;
        if (tid == 0) c_info[cchunk] = vinfo;
        return;
    }
    const uint4 s4 = *reinterpret_cast<const uint4*>(p_sdf + pb);
    const uint4 t4 = *reinterpret_cast<const uint4*>(p_type + pb);
    if (isect == 0ull) {  // wholly inside
        *reinterpret_cast<uint4*>(c_sdf + cb) = s4;
        *reinterpret_cast<uint4*>(c_type + cb) = t4;
        if (cp.extract) {
            *reinterpret_cast<uint4*>(p_sdf + pb) = void_sd;
            *reinterpret_cast<uint4*>(p_type + pb) = void_ty;
        }
        if (tid == 0) {
            ivx_chunk_info ci_ = vinfo;
            ci_.kind = ci_.gen_kind = pinfo.kind;
            ci_.uniform_type = pinfo.uniform_type;
            ci_.flags = pinfo.flags & CF_ONLY_EMPTY;
            c_info[cchunk] = ci_;
            if (cp.extract) p_info[pchunk] = vinfo;
        }
        return;
    }
    const uint32_t ti = tid >> 4, tj = tid & 15u;
    const float rx = (blo[0] + 0.5f) + (float)ti, ry = (blo[1] + 0.5f) + (float)tj, rz = (blo[2] + 0.5f) + 0.0f;
    float md[16];
    bool first = true;
    for (uint32_t q = 0; q < cp.n_planes; ++q) {
        if (!((isect >> q) & 1ull)) continue;
        const float* pl = cp.planes[q];
        const float base = plane_sd(pl, rx, ry, rz), step = pl[2];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float v = base + step * (float)k;
            md[k] = first ? v : fmaxf(md[k], v);
        }
        first = false;
    }
    uint32_t sw[4] = {s4.x, s4.y, s4.z, s4.w}, cw[4] = {0, 0, 0, 0}, pw[4] = {0, 0, 0, 0};
    bool poly_nonempty = false, poly_nonvoid = false, par_nonempty = false, par_nonvoid = false;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int orig = (int)(int8_t)((sw[k >> 2] >> (8 * (k & 3))) & 0xFFu);
        const int d = sd_clamped(md[k]);
        const int ps = orig > d ? orig : d;
        const int comp = sd_complement(d);
        const int qs = orig > comp ? orig : comp;
        poly_nonempty |= ps < 0;
        poly_nonvoid |= ps <= SD_VOID_LIMIT;
        par_nonempty |= qs < 0;
        par_nonvoid |= qs <= SD_VOID_LIMIT;
        cw[k >> 2] |= (uint32_t)(ps & 0xFF) << (8 * (k & 3));
        pw[k >> 2] |= (uint32_t)(qs & 0xFF) << (8 * (k & 3));
    }
    const int poly_ne = __syncthreads_or(poly_nonempty), poly_nv = __syncthreads_or(poly_nonvoid);
    const int par_ne = __syncthreads_or(par_nonempty), par_nv = __syncthreads_or(par_nonvoid);
    if (!poly_ne && !poly_nv) {
        *reinterpret_cast<uint4*>(c_sdf + cb) = void_sd;
        *reinterpret_cast<uint4*>(c_type + cb) = void_ty;
        if (tid == 0) c_info[cchunk] = vinfo;
    } else {
        *reinterpret_cast<uint4*>(c_sdf + cb) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
        *reinterpret_cast<uint4*>(c_type + cb) = t4;
        if (tid == 0) {
            ivx_chunk_info ci_ = vinfo;
            ci_.kind = ci_.gen_kind = KIND_NONUNIFORM;
            ci_.flags = poly_ne ? 0 : CF_ONLY_EMPTY;
            c_info[cchunk] = ci_;
        }
    }
    if (cp.extract) {
        if (!par_ne && !par_nv) {
            *reinterpret_cast<uint4*>(p_sdf + pb) = void_sd;
            *reinterpret_cast<uint4*>(p_type + pb) = void_ty;
            if (tid == 0) p_info[pchunk] = vinfo;
        } else {
            *reinterpret_cast<uint4*>(p_sdf + pb) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
            if (tid == 0) {
                ivx_chunk_info pi_ = pinfo;
                pi_.kind = pi_.gen_kind = KIND_NONUNIFORM;  // convert_to_non_uniform_if_uniform (object.rs:2530-2550)
                pi_.flags = par_ne ? 0 : CF_ONLY_EMPTY;
                pi_.uniform_type = 0;
                p_info[pchunk] = pi_;
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_clip(ClipParams cp, int8_t* __restrict__ p_sdf, uint8_t* __restrict__ p_type, ivx_chunk_info* __restrict__ p_info,
                                              int8_t* __restrict__ c_sdf, uint8_t* __restrict__ c_type, ivx_chunk_info* __restrict__ c_info) {
    clip_body(cp, p_sdf, p_type, p_info, c_sdf, c_type, c_info, blockIdx.x);
}
// All fragments of an impact in one launch (many.hpp): a block finds its fragment by the running block counts; the fragment's parameters —
// a kilobyte of planes — stay in memory and are read through the scalar cache as the body asks for them.
struct ClipManyArgs {
    ClipParams cp;
    int8_t* p_sdf;
    uint8_t* p_type;
    ivx_chunk_info* p_info;
    int8_t* c_sdf;
    uint8_t* c_type;
    ivx_chunk_info* c_info;
};
static_assert(sizeof(ClipManyArgs) % 8 == 0, "argument blocks travel as 8-byte words");
__global__ __launch_bounds__(256) void k_clip_many(const ClipManyArgs* __restrict__ argv, const uint32_t* __restrict__ block_end, uint32_t n) {
    const uint32_t i = ivx_many_find(block_end, n, blockIdx.x);
    const uint32_t b0 = i ? block_end[i - 1u] : 0u;
    const ClipManyArgs& a = argv[i];
    clip_body(a.cp, a.p_sdf, a.p_type, a.p_info, a.c_sdf, a.c_type, a.c_info, blockIdx.x - b0);
}
IVX_MANY_LAUNCHER(many_clip, k_clip_many, ClipManyArgs, 256)

}  // namespace

int ivx_launch_split_move(ivx_grid* parent, ivx_grid* child, const uint32_t lo[3], const uint32_t cc[3], uint32_t target) {
    ivx_planes_touched(parent);
    if (child) ivx_planes_touched(child);  // (no child: the region is discarded)
    SplitParams sp;
    sp.p = ivx_view(parent);
    for (int d = 0; d < 3; ++d) {
        sp.lo[d] = lo[d];
        sp.cc[d] = cc[d];
    }
    sp.target = target;
    {  // these kernels read and rewrite whole planes
        int rc = ivx_ensure_dense(parent);
        if (rc) return rc;
    }
    const uint32_t n = cc[0] * cc[1] * cc[2];
    IVX_KLAUNCH(k_split_move, dim3(n), dim3(256), 0, parent->ctx->stream, sp, parent->llabel, parent->rcompid, parent->sdf, parent->type,
                       parent->info, child ? child->sdf : nullptr, child ? child->type : nullptr, child ? child->info : nullptr);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_split_repack(ivx_grid* src, ivx_grid* dst, const uint32_t off[3]) {
    ivx_planes_touched(dst);
    {
        int rc = ivx_ensure_dense(src);
        if (rc) return rc;
    }
    IVX_KLAUNCH(k_split_repack, dim3(1), dim3(256), 0, src->ctx->stream, src->cc[0], src->cc[1], src->cc[2], off[0], off[1], off[2], src->sdf,
                       src->type, dst->sdf, dst->type, dst->info);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

static const int s_clip_many_registered = (ivx_many_register(IVX_MK_CLIP, many_clip, sizeof(ClipManyArgs)), 0);

int ivx_launch_clip(ivx_grid* parent, ivx_grid* child, const uint32_t lo[3], const uint32_t cc[3], const float* planes4, uint32_t n_planes, int extract) {
    ivx_planes_touched(parent);
    if (child) ivx_planes_touched(child);  // (no child: the region is discarded)
    ClipParams cp;
    cp.p = ivx_view(parent);
    for (int d = 0; d < 3; ++d) {
        cp.lo[d] = lo[d];
        cp.cc[d] = cc[d];
    }
    cp.n_planes = n_planes;
    cp.extract = extract;
    {
        int rc = ivx_ensure_dense(parent);
        if (rc) return rc;
    }
    memset(cp.planes, 0, sizeof(cp.planes));
    memcpy(cp.planes, planes4, sizeof(float) * 4 * n_planes);
    ClipManyArgs ma;
    memset(&ma, 0, sizeof(ma));
    ma.cp = cp;
    ma.p_sdf = parent->sdf, ma.p_type = parent->type, ma.p_info = parent->info;
    ma.c_sdf = child->sdf, ma.c_type = child->type, ma.c_info = child->info;
    // (recorded when the fragments of an impact are made together, ivx_copy_polyhedra: the clip belongs to the CHILD's chain)
    if (!extract && ivx_many_try(child->ctx, child, IVX_MK_CLIP, cc[0] * cc[1] * cc[2], ma)) return IVX_OK;
    IVX_KLAUNCH(k_clip, dim3(cc[0] * cc[1] * cc[2]), dim3(256), 0, parent->ctx->stream, cp, parent->sdf, parent->type, parent->info, child->sdf,
                       child->type, child->info);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
