// Microbenchmark: sustained fp32-input MFMA rate of the whole chip (v_mfma_f32_32x32x2_f32, 2 waves per SIMD, no memory
// traffic) with low-toggle and with random operands -- the practical ceiling the conv kernels should be priced against
// next to the nominal 157.3 TFLOP/s (the chip clocks to its power budget).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip ; run on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + i + 2048) & 4095]; }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u + 1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1], b[u], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1], b[u + 1], c3, 0, 0, 0);
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    float *d, *in;
    hipMalloc(&d, 1024 * 512 * 4);
    hipMalloc(&in, 4096 * 4);
    float h[4096];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant) {
        srand(1);
        for (int i = 0; i < 4096; ++i) h[i] = variant ? (float)rand() / RAND_MAX - 0.5f : 0.f;
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        for (int wgs = 256; wgs <= 1024; wgs *= 2) {
            const int iters = 100000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 0, 0, in, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)wgs * 8 * iters * 16 * 4096.0;
                if (rep) printf("%s operands, %4d WGs x 8 waves: %.2f ms  %.1f TFLOP/s  (=> %.2f GHz if the MFMA pipes never idle)\n",
                                variant ? "random" : "zero  ", wgs, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 2.4);
            }
        }
    }
    return 0;
}
