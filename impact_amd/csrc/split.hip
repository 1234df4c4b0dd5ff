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
    uint32_t in_region = 0, other = 0;  // bit k
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t lab = (lw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        if (lab != 255u) {
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
    const uint4 s4 = *reinterpret_cast<const uint4*>(p_sdf + pb);
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

}  // namespace

int ivx_launch_split_move(ivx_grid* parent, ivx_grid* child, const uint32_t lo[3], const uint32_t cc[3], uint32_t target) {
    SplitParams sp;
    sp.p = ivx_view(parent);
    for (int d = 0; d < 3; ++d) {
        sp.lo[d] = lo[d];
        sp.cc[d] = cc[d];
    }
    sp.target = target;
    const uint32_t n = cc[0] * cc[1] * cc[2];
    hipLaunchKernelGGL(k_split_move, dim3(n), dim3(256), 0, parent->ctx->stream, sp, parent->llabel, parent->rcompid, parent->sdf, parent->type,
                       parent->info, child ? child->sdf : nullptr, child ? child->type : nullptr, child ? child->info : nullptr);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_split_repack(ivx_grid* src, ivx_grid* dst, const uint32_t off[3]) {
    hipLaunchKernelGGL(k_split_repack, dim3(1), dim3(256), 0, src->ctx->stream, src->cc[0], src->cc[1], src->cc[2], off[0], off[1], off[2], src->sdf,
                       src->type, dst->sdf, dst->type, dst->info);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
