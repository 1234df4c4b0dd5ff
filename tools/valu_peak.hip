// VALU issue-rate probe for gfx950: how many wave64 VALU instructions per second does the chip retire with W waves per SIMD?
// (bench.py's `valu_roofline` needs the real ceiling: MI355X_MICROARCH.md says a wave64 v_fma_f32 holds a SIMD-32 for 2 cycles and
// that ONE wave alone issues one every 4.) Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {(float)threadIdx.x, 1.0f}, p1 = p0 + 1.0f, p2 = p0 + 2.0f, p3 = p0 + 3.0f, pa = {a, a}, pb = {b, b};
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {  // fma
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                             : "v"(a), "v"(b));
            } else if (MODE == 1) {  // mul + add pairs (what -ffp-contract=off code issues)
                asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                             "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                             : "v"(a), "v"(b));
            } else if (MODE == 3) {  // packed f32: two lanes' worth per instruction (counted as ONE wave-instruction below)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                             : "v"(pa), "v"(pb));
            } else if (MODE == 4) {  // single ops of the integer mix, one kind at a time would go here; this one: v_cndmask only
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                             : "v"(a), "v"(b));
            } else if (MODE == 5) {  // v_and / v_or / v_xor / v_add_u32 only
                asm volatile("v_and_b32 %0, %0, %8\n v_or_b32 %1, %1, %9\n v_xor_b32 %2, %2, %8\n v_add_u32 %3, %3, %9\n"
                             "v_and_b32 %4, %4, %8\n v_or_b32 %5, %5, %9\n v_xor_b32 %6, %6, %8\n v_add_u32 %7, %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                             : "v"(a), "v"(b));
            } else if (MODE == 6) {  // shifts / bit-field extracts
                asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_bfe_u32 %2, %2, 1, 30\n v_lshlrev_b32 %3, 1, %3\n"
                             "v_lshrrev_b32 %4, 1, %4\n v_bfe_u32 %5, %5, 1, 30\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                             : "v"(a), "v"(b));
            } else {  // integer ops (and/or/shift mix)
                asm volatile("v_and_b32 %0, %0, %8\n v_or_b32 %1, %1, %9\n v_lshlrev_b32 %2, 1, %2\n v_add_u32 %3, %3, %9\n"
                             "v_xor_b32 %4, %4, %8\n v_bfe_u32 %5, %5, 1, 30\n v_cndmask_b32 %6, %6, %8, vcc\n v_max_u32 %7, %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                             : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7)) + (p0.x + p1.y + p2.x + p3.y);
}

template <int MODE>
static void run(const char* name) {
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {  // 256 threads = 4 waves = one per SIMD, so wg_per_cu = waves per SIMD
        const int blocks = 256 * wg_per_cu;
        k_fma<MODE><<<blocks, 256>>>(d, 10, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_fma<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)blocks * 4 * iters * 64;
        printf("%s waves/SIMD %d: %.3f ms, %.1f G wave-instr/s (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, wg_per_cu, ms, winstr / ms / 1e6,
               1024.0 * 2.4e9 / (winstr / (ms * 1e-3)));
    }
    hipFree(d);
}

int main() {
    run<0>("fma    ");
    run<1>("mul+add");
    run<2>("integer");
    run<3>("packed ");
    run<4>("cndmask");
    run<5>("logic  ");
    run<6>("shifts ");
    return 0;
}
