// Microbenchmark: issue cost of v_mfma_f32_16x16x4_f32 in the persistent LSTM's pattern -- one wave per SIMD (256-thread workgroup, one per CU),
// NACC independent accumulators fed round robin, 128 MFMAs per "step" -- against the 32 cycles (8 passes) the instruction occupies the pipe.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_16x16x4_chain mfma_16x16x4_chain.hip ; run on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void chain_kernel(float* out, unsigned long long* cycles, int steps) {
    f32x4 acc[NACC];
    float a[16], b[16];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) { a[i] = 1e-3f * (float)(threadIdx.x + i); b[i] = 1e-3f * (float)(threadIdx.x * 3 + i); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int m = 0; m < 128; ++m)
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m & 15], b[(m * 7) & 15], acc[m % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 r = acc[0];
    for (int i = 1; i < NACC; ++i) r = r + acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NACC>
static void run(int wgs) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int steps = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chain_kernel<NACC>, dim3(wgs), dim3(256), 0, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain_kernel<NACC>, dim3(wgs), dim3(256), 0, 0, out, cyc, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost);
    const double per = (double)ms * 1e6 / steps / 128.0;   // ns per MFMA per wave
    printf("NACC=%d wgs=%3d: %.2f ns per MFMA (%.1f cycles at 2.4 GHz; s_memtime ticks per MFMA %.1f) -> 128 MFMAs = %.2f us\n", NACC, wgs, per, per * 2.4,
           (double)h[0] / steps / 128.0, per * 128 / 1e3);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int wgs : {1, 256}) { run<1>(wgs); run<2>(wgs); run<4>(wgs); run<8>(wgs); }
    return 0;
}
