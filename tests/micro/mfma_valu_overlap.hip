// Microbenchmark: does the fp32-input MFMA pipe overlap with fp32 VALU work issued by ANOTHER wave on the same SIMD?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip ; run on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode 0: every wave does MFMAs; 1: every wave does VALU FMAs; 2: waves 0-3 MFMA, waves 4-7 VALU (same SIMDs)
// mode 3: waves 0-3 MFMA, 4-7 idle; mode 4: waves 0-3 idle, 4-7 VALU
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int wid = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || ((mode == 2 || mode == 3) && wid < 4);
    const bool do_valu = mode == 1 || ((mode == 2 || mode == 4) && wid >= 4);
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    if (do_mfma) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (do_valu) {
        float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {       // 64 independent-ish FMAs per iteration
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.3f ms  (mfma/wave=%d, valu fma/wave=%d)\n", mode, ms, 4 * iters, 64 * iters);
        }
    }
    return 0;
}
