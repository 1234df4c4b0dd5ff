// a8 — mass / first moments / inertia moments of all non-empty voxels about the grid origin.
//
// Reference: VoxelObjectInertialPropertyManager::initialized_from
//   engine/crates/impact_voxel/src/object/inertia.rs:125-136, 615-699 (non-uniform), 703-754 (uniform),
//   756-790 (object sum). The reference integrates the unit cube of every voxel with f32 running
//   coordinates (xl = xh; xh += extent), which carries ~1e-3 relative rounding at 256^3. The cube
//   integrals are exact polynomials in the integer voxel index:
//       xh^2 - xl^2 = e^2 (2I+1)          xh^3 - xl^3 = e^3 (3I^2+3I+1)
//   so this kernel accumulates the integer forms in f64 (order-independent to ~1e-16) and applies the
//   e^3, e^4/2, e^5/3, e^5/4 factors once. Parity gate: 1e-5 relative against the f64 oracle.
//
// Sweep: NonUniform chunks (the active list) one workgroup each, one thread per (i,j) row, 2 B/voxel read (flags + type
// as two 16-byte loads per thread) into a per-chunk slot; Uniform chunks are closed forms; the slots and closed forms
// are summed in chunk order by k_inertia_sum and a fixed-order final launch (bitwise reproducible, no float atomics).
#include "chunk_passes.hpp"
#include "many.hpp"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Moments of the NonUniform chunks as a kernel of its own (the step path computes them inside k_derive's sweep): one workgroup
// per listed chunk, into the chunk's own slot, so the order of the final sum does not depend on the order of the list.
struct InertiaDenseArgs {
    GridView g;
    const uint8_t* flags;
    const float* dens;
    double* chunk_moments;
    const uint32_t* work_counts;
    const uint32_t* active_list;
    uint32_t x_off, pad_;
};
__device__ __forceinline__ void inertia_dense_body(const InertiaDenseArgs& a, uint32_t bid, uint32_t nb) {
    const GridView& g = a.g;
    const uint32_t x_off = a.x_off;
    const uint8_t* __restrict__ flags = a.flags;
    const float* __restrict__ dens = a.dens;
    double* __restrict__ chunk_moments = a.chunk_moments;
    const uint32_t* __restrict__ work_counts = a.work_counts;
    const uint32_t* __restrict__ active_list = a.active_list;
    __shared__ float s_dens[256];
    __shared__ double s_red[16][10];
    const uint32_t tid = threadIdx.x;
    s_dens[tid] = dens[tid];
    const int ti = tid >> 4, tj = tid & 15;
    const uint32_t n_active = work_counts[0];
    for (uint32_t li = ivx_xcd_remap(bid, nb); li < n_active; li += nb) {
        __syncthreads();  // s_dens ready / the previous chunk's s_red use is over
        const uint32_t entry = active_list[li];
        const uint32_t chunk = IVX_LIST_CHUNK(entry);
        if (IVX_LIST_KIND(entry) != KIND_NONUNIFORM) continue;  // Uniform chunks are closed forms in k_inertia_sum
        const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
        const size_t o = (size_t)chunk * IVX_CHUNK_VOXELS + (size_t)tid * 16;
        const uint4 f = *reinterpret_cast<const uint4*>(flags + o);
        const uint4 t = *reinterpret_cast<const uint4*>(g.type + o);
        const uint32_t fw[4] = {f.x, f.y, f.z, f.w}, tw[4] = {t.x, t.y, t.z, t.w};
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (!((fw[k >> 2] >> (8 * (k & 3))) & VF_EMPTY)) m |= 1u << k;
        chunk_moments_rows(tid, m, tw, s_dens, s_red, (ci + (int)x_off) * 16 + ti, cj * 16 + tj, ck * 16, chunk_moments + (size_t)chunk * 10);
    }
}
__global__ __launch_bounds__(256) void k_inertia_dense(InertiaDenseArgs a) { inertia_dense_body(a, blockIdx.x, gridDim.x); }
IVX_MANY_TWIN(k_inertia_dense_many, InertiaDenseArgs, inertia_dense_body, __launch_bounds__(256))
IVX_MANY_LAUNCHER(many_inertia_dense, k_inertia_dense_many, InertiaDenseArgs, 256)
static_assert(sizeof(InertiaDenseArgs) % 8 == 0, "argument blocks travel as 8-byte words");
static const int s_many_registered_inertia = (ivx_many_register(IVX_MK_INERTIA_DENSE, many_inertia_dense, sizeof(InertiaDenseArgs)), 0);

// Sum over chunks in chunk order, one THREAD per chunk: a Uniform chunk is 4096 voxels of one type
// (compute_moments_for_uniform_chunk, inertia.rs:703-754) and its moments are closed forms of its origin
// [sum_{X=X0}^{X0+15} (2X+1) = 32 X0 + 256; sum (3X^2+3X+1) = (X0+16)^3 - X0^3]; a NonUniform chunk contributes its slot.
__global__ __launch_bounds__(256) void k_inertia_sum(GridView g, uint32_t x_off, const float* __restrict__ dens,
                                                     const double* __restrict__ chunk_moments, double* __restrict__ partials) {
    __shared__ double s_red[4][10];
    const uint32_t tid = threadIdx.x;
    const uint32_t n_chunks = g.cx * g.cy * g.cz;
    double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t chunk = blockIdx.x * 256u + tid; chunk < n_chunks; chunk += gridDim.x * 256u) {
        const ivx_chunk_info ci_ = g.info[chunk];
        if (ci_.kind == KIND_UNIFORM) {
            const int ck = chunk % g.cz, cj = (chunk / g.cz) % g.cy, ci = chunk / (g.cz * g.cy);
            const double d = (double)dens[ci_.uniform_type];
            const double I0 = (double)((ci + (int)x_off) * 16), J0 = (double)(cj * 16), K0 = (double)(ck * 16);
            const double a1x = 32.0 * I0 + 256.0, a1y = 32.0 * J0 + 256.0, a1z = 32.0 * K0 + 256.0;
            const double I1 = I0 + 16.0, J1 = J0 + 16.0, K1 = K0 + 16.0;
            const double a2x = I1 * I1 * I1 - I0 * I0 * I0, a2y = J1 * J1 * J1 - J0 * J0 * J0, a2z = K1 * K1 * K1 - K0 * K0 * K0;
            s[0] += 4096.0 * d;
            s[1] += 256.0 * d * a1x;
            s[2] += 256.0 * d * a1y;
            s[3] += 256.0 * d * a1z;
            s[4] += 256.0 * d * (a2y + a2z);
            s[5] += 256.0 * d * (a2x + a2z);
            s[6] += 256.0 * d * (a2x + a2y);
            s[7] += 16.0 * d * a1x * a1y;
            s[8] += 16.0 * d * a1y * a1z;
            s[9] += 16.0 * d * a1x * a1z;
        } else if (ci_.kind == KIND_NONUNIFORM) {
#pragma unroll
            for (int m = 0; m < 10; ++m) s[m] += chunk_moments[(size_t)chunk * 10 + m];
        }
    }
    const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
    for (int m = 0; m < 10; ++m) {
        const double v = wave_sum(s[m]);
        if (lane == 0) s_red[wave][m] = v;
    }
    __syncthreads();
    if (tid < 10) partials[(size_t)blockIdx.x * 10 + tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
}

// fixed-order (bitwise reproducible) reduction of the per-block partials: wave q sums moment q — every lane a
// strided subset in index order, then a fixed shuffle tree
__global__ __launch_bounds__(640) void k_inertia_final(uint32_t n_blocks, float extent, const double* __restrict__ partials, double* __restrict__ out) {
    const uint32_t q = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    double s = 0.0;
    for (uint32_t b = lane; b < n_blocks; b += 64) s += partials[(size_t)b * 10 + q];
    s = wave_sum(s);
    if (lane == 0) {
        const double e = (double)extent, e2 = e * e, e3 = e2 * e, e4 = e2 * e2, e5 = e4 * e;
        const double f = q == 0 ? e3 : (q <= 3 ? 0.5 * e4 : (q <= 6 ? (1.0 / 3.0) * e5 : 0.25 * e5));
        out[q] = s * f;
    }
}

}  // namespace

static void launch_inertia_dense(ivx_grid* g, const GridView& v, const float* d_dens) {
    InertiaDenseArgs a;
    memset(&a, 0, sizeof(a));
    a.g = v, a.x_off = g->x_off, a.flags = g->flags, a.dens = d_dens, a.chunk_moments = g->chunk_moments, a.work_counts = ivx_wc(g), a.active_list = g->active_list;
    const uint32_t blocks = ivx_list_grid(g);
    if (!ivx_many_try(g->ctx, g, IVX_MK_INERTIA_DENSE, blocks, a)) IVX_KLAUNCH(k_inertia_dense, dim3(blocks), dim3(256), 0, g->ctx->stream, a);
}

int ivx_launch_inertia_dense(ivx_grid* g) {
    if (int rc_l = ivx_ensure_active_list(g)) return rc_l;
    GridView v = ivx_view(g);
    launch_inertia_dense(g, v, g->dens_dev);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}

int ivx_launch_inertia(ivx_grid* g, const float* d_dens, double* d_out10, int fused) {
    if (int rc_l = ivx_ensure_active_list(g)) return rc_l;
    uint32_t blocks = (g->n_chunks + 255u) / 256u;
    if (blocks > g->partial_blocks) blocks = (uint32_t)g->partial_blocks;
    GridView v = ivx_view(g);
    if (!fused)  // else k_derive left the chunk moments in the same sweep
        launch_inertia_dense(g, v, d_dens);
    IVX_KLAUNCH(k_inertia_sum, dim3(blocks), dim3(256), 0, g->ctx->stream, v, g->x_off, d_dens, g->chunk_moments, g->partials);
    IVX_KLAUNCH(k_inertia_final, dim3(1), dim3(640), 0, g->ctx->stream, blocks, g->extent, g->partials, d_out10);
    IVX_HIP_CHECK(hipGetLastError());
    return IVX_OK;
}
