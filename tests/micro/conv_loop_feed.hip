// Microbenchmark: the matrix waves' inner loop of conv_mfma_kernel<128,128,2,2> in isolation (no staging waves, no barriers, no global
// traffic): 2 x 2 register blocking of v_mfma_f32_32x32x2_f32, groups of two k-steps, the LDS reads of group g + 1 issued before the 8 MFMAs
// of group g.  Variants:  0 operands from registers (no LDS)          1 LDS reads with immediate offsets, addresses advanced by one add
//                         2 + the kernel's per-k-step address adds     3 + the k-step offset table (ds_read -> dependent address)
//                         4 = 3 with three register sets (reads two groups ahead)      5 = 3 with the reads issued between the MFMAs
// for 1 and 2 matrix waves per SIMD.  Tells how much of the 157 TFLOP/s the operand feed alone leaves.
// build: hipcc --offload-arch=gfx950 -O3 -o conv_loop_feed conv_loop_feed.hip ; run on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(512, 2) void loop_kernel(float* out, int iters, int nks2) {
    constexpr int BM = 128, TM = 2, TN = 2;
    __shared__ __attribute__((aligned(16))) float Ws[64 * BM];     // [k][m]
    __shared__ __attribute__((aligned(16))) float Xs[32 * 132 + 64];
    __shared__ int kofs_s[96];
    const int tid = threadIdx.x, lane = tid & 63, wid = (tid >> 6) & 3;
    for (int i = tid; i < 64 * BM; i += blockDim.x) Ws[i] = 1e-3f * (float)(i & 255);
    for (int i = tid; i < 32 * 132 + 64; i += blockDim.x) Xs[i] = 1e-3f * (float)((i * 7) & 255);
    for (int i = tid; i < 96; i += blockDim.x) kofs_s[i] = ((i >> 1) & 15) * 132 + (i & 1);       // channel row + tap, like a k = 2 layer
    __syncthreads();
    const int wm = wid / 2, wn = wid % 2, hi = lane >> 5, l31 = lane & 31;
    const float* WsA = Ws + hi * BM + wm * 64 + l31;
    const float* XbB = Xs + hi * 132 + wn * 64 + l31;
    const int2* kofs2 = (const int2*)kofs_s;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float ra[2][TM] = {{1.f, 2.f}, {3.f, 4.f}}, rb[2][TN] = {{.5f, .25f}, {.125f, 2.f}};
    auto load_group = [&](int g, const int2 k2, float (&a)[2][TM], float (&bb)[2][TN]) __attribute__((always_inline)) {
        if (V == 0) { for (int u = 0; u < 2; ++u) { for (int i = 0; i < TM; ++i) a[u][i] = ra[u][i]; for (int j = 0; j < TN; ++j) bb[u][j] = rb[u][j]; } return; }
        const int kos[2] = {k2.x, k2.y};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[u][i] = WsA[(g * 2 + u) * 2 * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[u][j] = V >= 3 ? XbB[kos[u] + j * 32] : (V == 2 ? XbB[((g * 2 + u) & 15) * 132 + j * 32] : XbB[u * 132 + j * 32]);
        }
    };
    auto mfma_group = [&](const float (&a)[2][TM], const float (&bb)[2][TN]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], bb[u][j], acc[i][j], 0, 0, 0);
    };
    if (V == 5) {     // the table variant, reads one group ahead, but ISSUED BETWEEN the MFMAs (one LDS / VALU instruction per MFMA shadow)
        for (int it = 0; it < iters; ++it) {
            float fa0[2][TM], fb0[2][TN], fa1[2][TM], fb1[2][TN];
            load_group(0, kofs2[0], fa0, fb0);
            int2 ko = kofs2[1];
            __builtin_amdgcn_sched_barrier(0);
            for (int g = 0; g < nks2; g += 2) {
                load_group(g + 1, ko, fa1, fb1);
                ko = kofs2[g + 2];
                mfma_group(fa0, fb0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                load_group(g + 2, ko, fa0, fb0);
                ko = kofs2[g + 3];
                mfma_group(fa1, fb1);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (V == 4) {     // the table variant with THREE register sets: the reads of group g + 2 are issued before the MFMAs of group g
        for (int it = 0; it < iters; ++it) {
            float fa0[2][TM], fb0[2][TN], fa1[2][TM], fb1[2][TN], fa2[2][TM], fb2[2][TN];
            load_group(0, kofs2[0], fa0, fb0);
            load_group(1, kofs2[1], fa1, fb1);
            int2 ko = kofs2[2];
            for (int g = 0; g < nks2; g += 3) {
                load_group(g + 2, ko, fa2, fb2); ko = kofs2[g + 3];
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                load_group(g + 3, ko, fa0, fb0); ko = kofs2[g + 4];
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                load_group(g + 4, ko, fa1, fb1); ko = kofs2[g + 5];
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(fa2, fb2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
        float fa0[2][TM], fb0[2][TN], fa1[2][TM], fb1[2][TN];
        load_group(0, kofs2[0], fa0, fb0);
        int2 ko = kofs2[1];
        for (int g = 0; g < nks2; g += 2) {
            load_group(g + 1, ko, fa1, fb1);
            if (V >= 3) ko = kofs2[g + 2];
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            load_group(g + 2, ko, fa0, fb0);
            if (V >= 3) ko = kofs2[g + 3];
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][7];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int V>
static void run(int threads) {
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    const int iters = 4000, nks2 = V == 4 ? 9 : 8, wgs = 256;     // ~16 k-steps per item like a 32-deep K chunk
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(loop_kernel<V>, dim3(wgs), dim3(threads), 0, 0, out, 10, nks2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop_kernel<V>, dim3(wgs), dim3(threads), 0, 0, out, iters, nks2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)wgs * (threads / 64) * iters * nks2 * 8.0;
    const double tf = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
    printf("variant %d, %d matrix waves per SIMD: %.1f TFLOP/s = %.3f of 157.3\n", V, threads / 256, tf, tf / 157.3);
    hipFree(out);
}
int main() {
    for (int threads : {256, 512}) { run<0>(threads); run<1>(threads); run<2>(threads); run<3>(threads); run<4>(threads); run<5>(threads); }
    return 0;
}
