// Microbenchmark / hardware probe (round 5): does global_load_lds_dwordx4 (LDS-DMA, 16 bytes per lane) place lane i's piece at
// M0-base + 16 i for ANY 16-byte aligned base, also when the 64 source addresses are contiguous?  Prints mismatches per LDS base offset.
// build: hipcc --offload-arch=gfx950 -O3 -o glds_align glds_align.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__global__ void probe(const float* __restrict__ src, float* __restrict__ out, int off_floats, int src_stride16) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8192; i += blockDim.x) smem[i] = -1.f;
    __syncthreads();
    float* dst = smem + off_floats + wid * 256;
    const float* g = src + (size_t)(wid * 64 + lane) * 4 * src_stride16;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)dst, 16, 0, 0);
    __syncthreads();
    for (int i = tid; i < 8192; i += blockDim.x) out[i] = smem[i];
}

int main() {
    const int N = 1 << 16;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *src, *out;
    hipMalloc(&src, N * 4); hipMalloc(&out, 8192 * 4);
    hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<float> o(8192);
    for (int stride : {1, 5, 8}) {
        for (int off : {0, 4, 8, 16, 32, 64, 128, 3076, 5124}) {
            if (off + 1024 > 8192) continue;
            hipLaunchKernelGGL(probe, dim3(1), dim3(256), 8192 * 4, 0, src, out, off, stride);
            hipMemcpy(o.data(), out, 8192 * 4, hipMemcpyDeviceToHost);
            int bad = 0, first = -1;
            for (int p = 0; p < 256; ++p)               // piece p of 4 floats: expected source floats (p * stride) * 4 .. + 3
                for (int s = 0; s < 4; ++s) {
                    const float want = (float)((p * stride) * 4 + s);
                    if (o[off + p * 4 + s] != want) { if (first < 0) first = p * 4 + s; ++bad; }
                }
            int stray = 0;
            for (int i = 0; i < 8192; ++i) if ((i < off || i >= off + 1024) && o[i] != -1.f) ++stray;
            printf("source stride %d x 16 B, LDS base offset %5d floats: %4d wrong of 1024 (first %d), %d stray writes; sample got %g %g %g %g want %g..\n",
                   stride, off, bad, first, stray, o[off + 4], o[off + 5], o[off + 260], o[off + 261], (float)(stride * 4));
        }
    }
    return 0;
}
