// Developer microbenchmark: a chain of ten small dependent kernels (what the voxel step's launches are to the queue), issued one by one on a
// stream and as one hipGraph launch: does the graph shorten the chain on the GPU? build: hipcc --offload-arch=gfx950 -O2 tools/graph_chain.hip -o /tmp/graph_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_small(unsigned* p, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1664525u + 1013904223u;
}
int main() {
    unsigned* d;
    const unsigned n = 1u << 15;
    hipMalloc(&d, n * 4);
    hipMemset(d, 0, n * 4);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int K = 10, REPS = 2000;
    auto run_stream = [&]() {
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s, d, n);
    };
    for (int r = 0; r < 50; ++r) run_stream();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REPS; ++r) {
        run_stream();
        hipStreamSynchronize(s);
    }
    auto t1 = std::chrono::steady_clock::now();
    printf("stream: %.2f us per chain of %d (with a host wait per chain)\n", 1e-3 * std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() / REPS, K);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REPS; ++r) run_stream();
    hipStreamSynchronize(s);
    t1 = std::chrono::steady_clock::now();
    printf("stream: %.2f us per chain, back to back (no wait between chains)\n", 1e-3 * std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() / REPS);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    run_stream();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 50; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REPS; ++r) {
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
    }
    t1 = std::chrono::steady_clock::now();
    printf("graph:  %.2f us per chain of %d (with a host wait per chain)\n", 1e-3 * std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() / REPS, K);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REPS; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    t1 = std::chrono::steady_clock::now();
    printf("graph:  %.2f us per chain, back to back\n", 1e-3 * std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() / REPS);
    return 0;
}
