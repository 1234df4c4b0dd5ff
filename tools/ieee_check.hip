// Checks that f32 sqrt and division in device code are IEEE correctly rounded (vs host SSE).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const float* a, const float* b, float* s1, float* s2, float* d1, float* d2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s1[i] = __fsqrt_rn(a[i]);
    s2[i] = sqrtf(a[i]);
    d1[i] = a[i] / b[i];
    d2[i] = __fdiv_rn(a[i], b[i]);
}
int main() {
    const int n = 1 << 20;
    std::vector<float> a(n), b(n), s1(n), s2(n), d1(n), d2(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        a[i] = (float)rand() / RAND_MAX * 100.0f;
        b[i] = (float)rand() / RAND_MAX * 10.0f + 0.001f;
    }
    float *da, *db, *ds1, *ds2, *dd1, *dd2;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&ds1, n * 4); hipMalloc(&ds2, n * 4); hipMalloc(&dd1, n * 4); hipMalloc(&dd2, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, ds1, ds2, dd1, dd2, n);
    hipMemcpy(s1.data(), ds1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s2.data(), ds2, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(d1.data(), dd1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(d2.data(), dd2, n * 4, hipMemcpyDeviceToHost);
    int e[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        volatile float hs = std::sqrt(a[i]);
        volatile float hd = a[i] / b[i];
        e[0] += s1[i] != hs; e[1] += s2[i] != hs; e[2] += d1[i] != hd; e[3] += d2[i] != hd;
    }
    printf("mismatches: __fsqrt_rn %d  sqrtf %d  operator/ %d  __fdiv_rn %d  of %d\n", e[0], e[1], e[2], e[3], n);
    return 0;
}
