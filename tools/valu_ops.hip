// Issue cost of single VALU opcodes on gfx950 (developer probe): 8 independent dependency chains per lane, 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/valu_ops.hip -o /tmp/valu_ops && /tmp/valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(OP, TAIL)                                                                                                                     \
    asm volatile(OP " %0, %0, " TAIL "\n" OP " %1, %1, " TAIL "\n" OP " %2, %2, " TAIL "\n" OP " %3, %3, " TAIL "\n" OP " %4, %4, " TAIL "\n" OP " %5, %5, " TAIL \
                    "\n" OP " %6, %6, " TAIL "\n" OP " %7, %7, " TAIL "\n"                                                                   \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)                                           \
                 : "v"(a), "v"(b), "s"(m))

#define UNARY8(OP)                                                                                                                      \
    asm volatile(OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b, unsigned long long m) {
    float t0 = 0, t1 = 0;
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    asm volatile("s_mov_b64 vcc, %0" ::"s"(m) : "vcc");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) CHAIN8("v_cndmask_b32_e32", "%8, vcc");
            if (MODE == 1) CHAIN8("v_cndmask_b32_e64", "%8, %10");
            if (MODE == 2) CHAIN8("v_bfi_b32", "%8, %9");
            if (MODE == 3) CHAIN8("v_add_f32", "%8");
            if (MODE == 4) CHAIN8("v_mul_lo_u32", "%8");
            if (MODE == 5) CHAIN8("v_mad_u32_u24", "%8, %9");
            if (MODE == 6) CHAIN8("v_lshlrev_b32", "%8");
            if (MODE == 7) CHAIN8("v_and_or_b32", "%8, %9");
            if (MODE == 8) CHAIN8("v_perm_b32", "%8, %9");
            if (MODE == 9) CHAIN8("v_max3_f32", "%8, %9");
            if (MODE == 10) CHAIN8("v_add3_u32", "%8, %9");
            if (MODE == 11) CHAIN8("v_bfe_u32", "%8, %9");
            if (MODE == 12) CHAIN8("v_mul_f32", "%8");
            if (MODE == 13) CHAIN8("v_min_u32", "%8");
            if (MODE == 30) UNARY8("v_sqrt_f32");
            if (MODE == 31) UNARY8("v_rcp_f32");
            if (MODE == 32) UNARY8("v_rsq_f32");
            if (MODE == 33) UNARY8("v_cvt_i32_f32");
            if (MODE == 34) CHAIN8("v_fma_f32", "%8, %9");
            if (MODE == 35) {  // the shape of a correctly rounded square root: one v_sqrt_f32 and eight plain operations, two chains (18 per 2)
                asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n"
                             "v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                             "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                             "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
            if (MODE == 20) {  // compare + select through vcc, 8 independent selects
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32_e32 %1, %1, %9, vcc\n"
                             "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32_e32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32_e32 %3, %3, %9, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32_e32 %4, %4, %9, vcc\n v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32_e32 %5, %5, %9, vcc\n"
                             "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32_e32 %6, %6, %9, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32_e32 %7, %7, %9, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(m) : "vcc");
            }
            if (MODE == 21) {  // compare into an SGPR pair + select (e64), 8 independent selects
                asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cndmask_b32_e64 %1, %1, %9, s[22:23]\n"
                             "v_cmp_lt_f32 s[24:25], %2, %8\n v_cndmask_b32_e64 %2, %2, %9, s[24:25]\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cndmask_b32_e64 %3, %3, %9, s[26:27]\n"
                             "v_cmp_lt_f32 s[20:21], %4, %8\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cndmask_b32_e64 %5, %5, %9, s[22:23]\n"
                             "v_cmp_lt_f32 s[24:25], %6, %8\n v_cndmask_b32_e64 %6, %6, %9, s[24:25]\n v_cmp_lt_f32 s[26:27], %7, %8\n v_cndmask_b32_e64 %7, %7, %9, s[26:27]\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(m)
                             : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            }
            if (MODE == 23) {  // ONE compare, four selects on it through vcc (x2 groups)
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %9, vcc\n v_cndmask_b32_e32 %1, %1, %9, vcc\n v_cndmask_b32_e32 %2, %2, %9, vcc\n v_cndmask_b32_e32 %3, %3, %9, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32_e32 %4, %4, %9, vcc\n v_cndmask_b32_e32 %5, %5, %9, vcc\n v_cndmask_b32_e32 %6, %6, %9, vcc\n v_cndmask_b32_e32 %7, %7, %9, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(m) : "vcc");
            }
            if (MODE == 24) {  // ONE compare into an SGPR pair, four selects on it (x2 groups)
                asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cndmask_b32_e64 %1, %1, %9, s[20:21]\n v_cndmask_b32_e64 %2, %2, %9, s[20:21]\n v_cndmask_b32_e64 %3, %3, %9, s[20:21]\n"
                             "v_cmp_lt_f32 s[22:23], %4, %8\n v_cndmask_b32_e64 %4, %4, %9, s[22:23]\n v_cndmask_b32_e64 %5, %5, %9, s[22:23]\n v_cndmask_b32_e64 %6, %6, %9, s[22:23]\n v_cndmask_b32_e64 %7, %7, %9, s[22:23]\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(m)
                             : "s20", "s21", "s22", "s23");
            }
            if (MODE == 22) {  // the same select by integer arithmetic: difference of the (non-negative) bit patterns, sign smeared, bit-field insert
                asm volatile("v_sub_u32 %10, %0, %8\n v_ashrrev_i32 %10, 31, %10\n v_bfi_b32 %0, %10, %9, %0\n v_sub_u32 %11, %1, %8\n v_ashrrev_i32 %11, 31, %11\n v_bfi_b32 %1, %11, %9, %1\n"
                             "v_sub_u32 %10, %2, %8\n v_ashrrev_i32 %10, 31, %10\n v_bfi_b32 %2, %10, %9, %2\n v_sub_u32 %11, %3, %8\n v_ashrrev_i32 %11, 31, %11\n v_bfi_b32 %3, %11, %9, %3\n"
                             "v_sub_u32 %10, %4, %8\n v_ashrrev_i32 %10, 31, %10\n v_bfi_b32 %4, %10, %9, %4\n v_sub_u32 %11, %5, %8\n v_ashrrev_i32 %11, 31, %11\n v_bfi_b32 %5, %11, %9, %5\n"
                             "v_sub_u32 %10, %6, %8\n v_ashrrev_i32 %10, 31, %10\n v_bfi_b32 %6, %10, %9, %6\n v_sub_u32 %11, %7, %8\n v_ashrrev_i32 %11, 31, %11\n v_bfi_b32 %7, %11, %9, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(t0), "v"(t1));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

template <int MODE>
static void run(const char* name) {
    float* d;
    (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 4;
    k<MODE><<<blocks, 256>>>(d, 10, 1.0001f, 0.5f, 0x5555555555555555ull);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f, 0x5555555555555555ull);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 64;
    printf("%-22s %.3f ms  %.0f G wave-instr/s\n", name, ms, winstr / ms / 1e6);
    (void)hipFree(d);
}

int main() {
    run<3>("v_add_f32");
    run<12>("v_mul_f32");
    run<0>("v_cndmask_b32 vcc");
    run<1>("v_cndmask_b32 sgpr");
    run<2>("v_bfi_b32");
    run<4>("v_mul_lo_u32");
    run<5>("v_mad_u32_u24");
    run<6>("v_lshlrev_b32");
    run<11>("v_bfe_u32");
    run<7>("v_and_or_b32");
    run<8>("v_perm_b32");
    run<9>("v_max3_f32");
    run<10>("v_add3_u32");
    run<13>("v_min_u32");
    run<34>("v_fma_f32");
    run<30>("v_sqrt_f32");
    run<31>("v_rcp_f32");
    run<32>("v_rsq_f32");
    run<33>("v_cvt_i32_f32");
    run<35>("2 sqrt + 16 plain (x18/8)");
    run<20>("cmp+cndmask vcc (x2)");
    run<21>("cmp+cndmask sgpr (x2)");
    run<22>("sub+ashr+bfi (x3)");
    run<23>("cmp + 4 cndmask vcc (10 per 8)");
    run<24>("cmp + 4 cndmask sgpr (10 per 8)");
    return 0;
}
