// Developer tool: how v_sqrt_f32 (the hardware estimate under sqrt_rn, sdf_sample.hip) errs against the correctly rounded root, over EVERY
// positive normal f32: equal / one ulp low / one ulp high / further off — and whether a one-sided fix-up would do.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O2 tools/sqrt_bias.hip -o /tmp/sqrt_bias && /tmp/sqrt_bias
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long* cnt, uint32_t first, uint32_t* worst) {
    const uint32_t bits = first + blockIdx.x * 256u + threadIdx.x;
    if (bits < 0x00800000u || bits >= 0x7F800000u) return;
    const float x = __uint_as_float(bits);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float c = (float)sqrt((double)x);  // correctly rounded (the double's 53 bits cannot sit on a float's midpoint for a root)
    const int d = (int)(__float_as_uint(s) - __float_as_uint(c));
    int slot = d == 0 ? 0 : (d == -1 ? 1 : (d == 1 ? 2 : 3));
    // the one-sided candidates: pick the neighbour above when its product with s stays below x
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    const float fix_up = r_up > 0.0f ? s_up : s;
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float fix_dn = r_dn <= 0.0f ? s_dn : s;
    atomicAdd(&cnt[slot], 1ull);
    if (fix_up != c) atomicAdd(&cnt[4], 1ull);
    if (fix_dn != c) atomicAdd(&cnt[5], 1ull);
    if (slot == 3) atomicMax(worst, (uint32_t)(d < 0 ? -d : d));
}
int main() {
    unsigned long long* d_cnt;
    uint32_t* d_worst;
    hipMalloc(&d_cnt, 8 * 8);
    hipMalloc(&d_worst, 4);
    hipMemset(d_cnt, 0, 64);
    hipMemset(d_worst, 0, 4);
    for (uint32_t first = 0x00800000u; first < 0x7F800000u; first += 1u << 28) hipLaunchKernelGGL(k, dim3((1u << 28) / 256u), dim3(256), 0, 0, d_cnt, first, d_worst);
    unsigned long long c[8];
    uint32_t w;
    hipMemcpy(c, d_cnt, 64, hipMemcpyDeviceToHost);
    hipMemcpy(&w, d_worst, 4, hipMemcpyDeviceToHost);
    printf("v_sqrt_f32 over all positive normal f32: equal %llu, one ulp low %llu, one ulp high %llu, further off %llu (worst %u ulp)\n", c[0], c[1], c[2], c[3], w);
    printf("fix-up with the upper neighbour alone wrong for %llu inputs; with the lower neighbour alone wrong for %llu\n", c[4], c[5]);
    return 0;
}
