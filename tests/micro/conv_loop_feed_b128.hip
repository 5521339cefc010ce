// Microbenchmark (round 5): the matrix waves' inner loop with a QUAD-K operand layout -- one ds_read_b128 per operand tile per FOUR k-steps
// instead of one ds_read_b32 per k-step (conv_loop_feed.hip measures the round 1-4 form: 0.80 / 0.89 of the fp32 MFMA peak with one / two
// matrix waves per SIMD against 0.955 / 0.995 from registers).
//   A (weights, packed offline):  Ws[kq][hi][BM rows][4]      lane (row, hi) reads its 4 k values of quad kq as one 16-byte piece
//   B (input slab):               Xs[c8][hi][columns][4 ch]   lane (column, hi) reads channels 8 c8 + 4 hi + 0..3 of its column; a tap is a
//                                                             column offset (16 bytes per column), so a quad = (tap, 8 channels)
// k-step s of a quad multiplies channel 8 c8 + s (hi = 0) and 8 c8 + 4 + s (hi = 1): A and B agree on that order, which is all the GEMM needs.
// Variants: 0 = round-4 form (b32 reads + offset table, for the same-box baseline)   1 = quad layout, B offsets from a table in LDS
//           2 = quad layout, offsets computed (no table read)                        3 = quad layout, reads TWO quads ahead
// each for 1, 2 (and, LDS permitting, 3 / 4) matrix waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o conv_loop_feed_b128 conv_loop_feed_b128.hip ; run on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, TM = 2, TN = 2, COLS = 132, NQ = 8;      // 8 quads per item = 2 taps x 32 channels (a CC = 32 chunk of a 2-tap GEMM)

template <int V, int WPS>
__global__ __launch_bounds__(256, WPS) void loop_kernel(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float Ws[(V == 0 ? NQ : 4) * 2 * BM * 4];       // 32 KiB (quad variants: 16 KiB, quads reuse 4 images)
    __shared__ __attribute__((aligned(16))) float Xs[4 * 2 * COLS * 4 + 64];
    __shared__ int kofs_s[64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < (V == 0 ? NQ : 4) * 2 * BM * 4; i += blockDim.x) Ws[i] = 1e-3f * (float)(i & 255);
    for (int i = tid; i < 4 * 2 * COLS * 4 + 64; i += blockDim.x) Xs[i] = 1e-3f * (float)((i * 7) & 255);
    for (int i = tid; i < 64; i += blockDim.x) kofs_s[i] = V == 0 ? ((i >> 1) & 15) * COLS + (i & 1) : ((i & 3) * (2 * COLS * 4) + ((i >> 2) & 1) * 4);
    __syncthreads();
    const int wm = wid / 2, wn = wid % 2, hi = lane >> 5, l31 = lane & 31;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (V == 0) {
        const float* WsA = Ws + hi * BM + wm * 64 + l31;
        const float* XbB = Xs + hi * COLS + wn * 64 + l31;
        const int2* kofs2 = (const int2*)kofs_s;
        auto load_group = [&](int g, const int2 k2, float (&a)[2][TM], float (&bb)[2][TN]) __attribute__((always_inline)) {
            const int kos[2] = {k2.x, k2.y};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[u][i] = WsA[(g * 2 + u) * 2 * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[u][j] = XbB[kos[u] + j * 32];
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
        const int nks2 = NQ * 2;                    // groups of two k-steps: the same 32 k-steps per item
        for (int it = 0; it < iters; ++it) {
            float fa0[2][TM], fb0[2][TN], fa1[2][TM], fb1[2][TN];
            load_group(0, kofs2[0], fa0, fb0);
            int2 ko = kofs2[1];
            for (int g = 0; g < nks2; g += 2) {
                load_group(g + 1, ko, fa1, fb1);
                ko = kofs2[(g + 2) & 15];
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                load_group((g + 2) & 15, ko, fa0, fb0);
                ko = kofs2[(g + 3) & 15];
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        const f32x4* WsA = (const f32x4*)Ws + hi * BM + wm * 64 + l31;                 // + kq * 2 * BM + i * 32
        const char* XbB = (const char*)Xs + 16 * (hi * COLS + wn * 64 + l31);          // + byte offset of the quad + j * 32 * 16
        auto load_quad = [&](int q, int bofs, f32x4 (&a)[TM], f32x4 (&bb)[TN]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = WsA[(q & 3) * 2 * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[j] = *(const f32x4*)(XbB + 4 * bofs + j * 32 * 16);
        };
        auto mfma_quad = [&](const f32x4 (&a)[TM], const f32x4 (&bb)[TN]) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], bb[j][s], acc[i][j], 0, 0, 0);
        };
        auto ofs = [&](int q) __attribute__((always_inline)) { return V == 1 ? kofs_s[q] : (q & 3) * (2 * COLS * 4) + ((q >> 2) & 1) * 4; };
        if (V == 3) {
            for (int it = 0; it < iters; ++it) {
                f32x4 a0[TM], b0[TN], a1[TM], b1[TN], a2[TM], b2[TN];
                load_quad(0, ofs(0), a0, b0);
                load_quad(1, ofs(1), a1, b1);
#pragma unroll
                for (int q = 0; q < NQ - 2; q += 3) {          // 0..5, then the tail below
                    load_quad(q + 2, ofs(q + 2), a2, b2);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_quad(a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    load_quad((q + 3) & 7, ofs((q + 3) & 7), a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_quad(a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                    load_quad((q + 4) & 7, ofs((q + 4) & 7), a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_quad(a2, b2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfma_quad(a0, b0);
                mfma_quad(a1, b1);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
                load_quad(0, ofs(0), a0, b0);
#pragma unroll
                for (int q = 0; q < NQ; q += 2) {
                    load_quad(q + 1, ofs(q + 1), a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_quad(a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    load_quad((q + 2) & 7, ofs((q + 2) & 7), a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_quad(a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][7];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int V, int WPS>
static void run() {
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    const int iters = 4000, wgs = 256 * WPS;          // WPS workgroups of 4 waves per CU = WPS matrix waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((loop_kernel<V, WPS>), dim3(wgs), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop_kernel<V, WPS>), dim3(wgs), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)wgs * 4 * iters * NQ * 16.0;
    const double tf = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, loop_kernel<V, WPS>, 256, 0);
    printf("variant %d, %d matrix waves per SIMD (occupancy %d wg/CU): %.1f TFLOP/s = %.3f of 157.3\n", V, WPS, occ, tf, tf / 157.3);
    hipFree(out);
}
int main() {
    run<0, 1>(); run<1, 1>(); run<2, 1>(); run<3, 1>();
    run<0, 2>(); run<1, 2>(); run<2, 2>(); run<3, 2>();
    run<0, 3>(); run<1, 3>(); run<2, 3>();
    run<0, 4>(); run<1, 4>(); run<2, 4>();
    return 0;
}
