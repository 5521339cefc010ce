// Microbenchmark: cost of a chain of DEPENDENT small kernels on one stream (the shape of a batch-8 decoding step: 62 kernels, each waiting
// for its predecessor), eager launches vs one captured HIP graph, for an empty kernel and for a kernel that hands 16 KiB through global
// memory like the step's GEMVs do.  build: hipcc --offload-arch=gfx950 -O3 -o dependent_launch dependent_launch.hip ; run on MI355X
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 0.f; }

// every workgroup reads the whole 16 KiB vector its predecessor wrote and writes its own slice of the next one
__global__ void relay_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        for (int j = 0; j < 32; ++j) out[blockIdx.x * 32 + j] = t * 1e-9f + (float)j;
    }
}

template <typename F>
static double time_chain(hipStream_t st, int chain, int reps, bool graph, F launch) {
    hipGraphExec_t exec = nullptr;
    if (graph) {
        hipGraph_t g;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < chain; ++i) launch(i);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
    }
    auto run = [&]() { if (graph) hipGraphLaunch(exec, st); else for (int i = 0; i < chain; ++i) launch(i); };
    for (int r = 0; r < 3; ++r) run();
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) run();
    hipStreamSynchronize(st);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (exec) hipGraphExecDestroy(exec);
    return us / reps / chain;
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    float *a, *b;
    hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20);
    hipMemset(a, 0, 1 << 20); hipMemset(b, 0, 1 << 20);
    const int chain = 62, reps = 200;
    for (int graph = 0; graph < 2; ++graph) {
        for (int wgs : {1, 96}) for (int threads : {64, 1024}) {
            const double us = time_chain(st, chain, reps, graph, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(threads), 0, st, a); });
            printf("%s empty kernel   %3d WGs x %4d threads: %.2f us per dependent kernel\n", graph ? "graph" : "eager", wgs, threads, us);
        }
        for (int wgs : {32, 96, 128}) for (int threads : {256, 1024}) {
            const double us = time_chain(st, chain, reps, graph, [&](int i) {
                hipLaunchKernelGGL(relay_kernel, dim3(wgs), dim3(threads), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, 4096); });
            printf("%s relay 16 KiB   %3d WGs x %4d threads: %.2f us per dependent kernel\n", graph ? "graph" : "eager", wgs, threads, us);
        }
    }
    return 0;
}
