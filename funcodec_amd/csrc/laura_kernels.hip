// gfx950 (MI355X / CDNA4) kernels of the LauraTTS generation path (laura_kernels.h).  fp32 everywhere; contractions on the
// fp32-input matrix cores (v_mfma_f32_16x16x4_f32: an fmaf chain in k order, no narrower arithmetic).  Wavefront = 64 lanes.
//
// MFMA 16x16x4 operand layout (as used by the codec kernels): lane l = 16 g + r;  A operand = A[row r][k g];
// B operand = B[k g][col r];  accumulator register v = D[row 4 g + v][col r].
#include "laura_kernels.h"
#include "kernels.h"

#include <atomic>
#include <cstdlib>

namespace fc {
namespace laura {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

namespace {

__device__ __forceinline__ int ceil_div_d(int a, int b) { return (a + b - 1) / b; }
inline int ceil_div_h(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float act_f(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v / (1.f + expf(-v));      // Swish: x * sigmoid(x) (nets_utils.py:568-574)
    return v;
}

// opt-in to > 64 KiB of dynamic LDS, once per (kernel, device)
template <typename F>
hipError_t big_lds(F kfn, std::atomic<unsigned long long>& done, int bytes = 160 * 1024) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    return hipSuccess;
}

}  // namespace

// =====================================================================================================================
// LayerNorm over the channel axis of a feature-major tensor (torch.nn.LayerNorm, biased variance; eps 1e-12 inside the blocks,
// funcodec/modules/layer_norm.py:22, 1e-5 in the input layers).  One thread column per time step (coalesced over t), the C
// channels split over 4 thread rows; two-pass mean / variance in fp32.
// =====================================================================================================================
// round 3: a workgroup = 16 time steps x 16 channel groups; a thread keeps its C / 16 values in registers (one read of the tensor, one
// write) and the statistics meet in LDS.  The first version (64 time steps x 4 channel groups, three passes over global memory) ran
// 56 workgroups for a batch of 8 x 430 columns and took ~100 us per call.
template <int CPT>                                     // channels per thread = C / 16 (<= CPT)
__global__ __launch_bounds__(256) void layernorm_fm_kernel(const float* __restrict__ x, const float* __restrict__ add, float* sum_out,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           int relu, float post_scale, float* __restrict__ y, int C, int T) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, cg = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + tx, b = blockIdx.y;
    const bool ok = t < T;
    const size_t base = (size_t)b * C * T + (ok ? t : 0);
    const int cpt = C / 16, c0 = cg * cpt;
    float v[CPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        v[i] = 0.f;
        if (i < cpt) {
            float u = x[base + (size_t)(c0 + i) * T];
            if (add) u += add[base + (size_t)(c0 + i) * T];
            v[i] = u;
            s += u;
        }
    }
    red[cg][tx] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) tot += red[g][tx];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (i < cpt) { const float dv = v[i] - mean; q += dv * dv; }
    red[cg][tx] = q;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) tot += red[g][tx];
    const float rstd = 1.f / sqrtf(tot / (float)C + eps);
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (i < cpt) {
            const int c = c0 + i;
            if (sum_out) sum_out[base + (size_t)c * T] = v[i];
            float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
            if (relu) o = o > 0.f ? o : 0.f;
            y[base + (size_t)c * T] = o * post_scale;
        }
}

hipError_t launch_layernorm_fm(const float* x, const float* add, float* sum_out, const float* gamma, const float* beta, float eps,
                               int relu, float post_scale, float* y, int B, int C, int T, hipStream_t st) {
    if (C % 16 || C > 1024) return hipErrorInvalidValue;
    const dim3 grid(ceil_div_h(T, 16), B);
    if (C <= 128) hipLaunchKernelGGL(layernorm_fm_kernel<8>, grid, dim3(256), 0, st, x, add, sum_out, gamma, beta, eps, relu, post_scale, y, C, T);
    else if (C <= 512) hipLaunchKernelGGL(layernorm_fm_kernel<32>, grid, dim3(256), 0, st, x, add, sum_out, gamma, beta, eps, relu, post_scale, y, C, T);
    else hipLaunchKernelGGL(layernorm_fm_kernel<64>, grid, dim3(256), 0, st, x, add, sum_out, gamma, beta, eps, relu, post_scale, y, C, T);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void act_kernel(float* x, size_t n, int act) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        f32x4 v = *(f32x4*)(x + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_f(v[j], act);
        *(f32x4*)(x + i) = v;
    } else {
        for (size_t k = i; k < n; ++k) x[k] = act_f(x[k], act);
    }
}
hipError_t launch_act(float* x, size_t n, int act, hipStream_t st) {
    if (!act || !n) return hipSuccess;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, x, n, act);
    return hipGetLastError();
}

// =====================================================================================================================
// Full-sequence rel-pos attention (RelPositionMultiHeadedAttention.forward, funcodec/modules/attention.py:265-308 +
// forward_attention :64-96).  One workgroup = 16 queries of one (utterance, head):
//   A  scores of the 16 queries against every key tile (4 waves, one 16-key tile each per trip) into LDS:
//        ac = (q + u) . k_j                       one 16x16 MFMA tile
//        bd = (q + v) . p_{i-j}                   the 31 relative positions of the tile as two MFMA tiles against the position
//                                                 table, the (i, j) diagonal picked through LDS (= the reference's rel_shift)
//        s  = (ac + bd) / sqrt(dk), masked (padding; causal with a bidirectional prefix block for the LM)
//   B  row softmax in LDS (max, exp, sum, divide; masked entries 0 like masked_fill(mask, 0.0))
//   C  ctx = P . V: wave w owns 16 of the dk output dims, A operand = probabilities from LDS, B operand = V rows (16-byte loads)
// =====================================================================================================================
struct AttnFullArgs {
    const float *qkv, *ptab, *bias_u, *bias_v;
    const int *lens, *bidir;
    float* ctx;
    int causal, H, T, R, PR, SW;     // SW = LDS row stride of the score block
};

template <int DK>
__global__ __launch_bounds__(256) void attn_full_kernel(AttnFullArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* S = lds;                          // [16][SW]
    float* Gs = lds + 16 * a.SW;             // [4 waves][16][33]
    constexpr int NC = DK / 16;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const int i0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const int T = a.T, d = a.H * DK;
    const int len = a.lens[b];
    const int bid = (a.causal && a.bidir) ? a.bidir[b] : 0;
    if (i0 >= len) {       // padded queries: the reference computes garbage there too and every consumer masks them; write zeros
        if (w < NC) {
            const int t0 = i0 + 4 * g;
            if (t0 < T) *(f32x4*)(a.ctx + ((size_t)b * d + h * DK + 16 * w + r16) * T + t0) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    // key tiles this query tile can see at all
    int nkt = ceil_div_d(len, 16);
    if (a.causal) {
        int lim = i0 + 16;
        if (bid > lim) lim = bid;
        if (lim < len) nkt = ceil_div_d(lim, 16);
    }
    const float* qb = a.qkv + (size_t)b * 3 * d * T;
    const float* kb = qb + (size_t)d * T;
    const float* vb = kb + (size_t)d * T;
    // query fragments (A operand): lane (r16 = query, g) holds dims 16 c + 4 g + j
    const int iq = (i0 + r16 < T) ? i0 + r16 : T - 1;
    float qu[NC][4], qv[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dd = h * DK + 16 * c + 4 * g + j;
            const float q = qb[(size_t)dd * T + iq];
            qu[c][j] = q + a.bias_u[dd];
            qv[c][j] = q + a.bias_v[dd];
        }
    const float scale = 1.f / sqrtf((float)DK);
    const float NEG = -3.4028234663852886e38f;       // numpy.finfo(float32).min (attention.py:79-82)
    float* Gw = Gs + w * 16 * 33;
    const int ntrip = ceil_div_d(nkt, 4);
    for (int trip = 0; trip < ntrip; ++trip) {
        const int kt = trip * 4 + w;
        const bool active = kt < nkt;
        const int j0 = kt * 16;
        f32x4 ac = {0.f, 0.f, 0.f, 0.f}, g0 = ac, g1 = ac;
        if (active) {
            const int jk = (j0 + r16 < T) ? j0 + r16 : T - 1;
            const int rlo = i0 - j0 - 15 + (a.R - 1);                 // table column of window column 0
            int p0 = rlo + r16, p1 = rlo + 16 + r16;
            p0 = p0 < 0 ? 0 : (p0 >= a.PR ? a.PR - 1 : p0);
            p1 = p1 < 0 ? 0 : (p1 >= a.PR ? a.PR - 1 : p1);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int dd = h * DK + 16 * c + 4 * g + j;
                    const float kv = kb[(size_t)dd * T + jk];
                    const float pa = a.ptab[(size_t)dd * a.PR + p0];
                    const float pb = a.ptab[(size_t)dd * a.PR + p1];
                    ac = __builtin_amdgcn_mfma_f32_16x16x4f32(qu[c][j], kv, ac, 0, 0, 0);
                    g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[c][j], pa, g0, 0, 0, 0);
                    g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[c][j], pb, g1, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Gw[(4 * g + r) * 33 + r16] = g0[r];
                Gw[(4 * g + r) * 33 + 16 + r16] = g1[r];
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = 4 * g + r, i = i0 + il, j = j0 + r16;
                const float bd = Gw[il * 33 + (il - r16 + 15)];      // window column of relative position i - j
                bool vis = j < len;
                if (a.causal) vis = vis && (j <= i || (i < bid && j < bid));
                S[il * a.SW + j] = vis ? (ac[r] + bd) * scale : NEG;
            }
        }
        __syncthreads();
    }
    // ---- B: softmax, 16 threads per row
    {
        const int row = tid >> 4, part = tid & 15;
        const int nk = nkt * 16;
        float* Sr = S + row * a.SW;
        float m = NEG;
        for (int j = part; j < nk; j += 16) m = fmaxf(m, Sr[j]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float s = 0.f;
        for (int j = part; j < nk; j += 16) {
            const float v = Sr[j];
            const float e = (v == NEG) ? 0.f : expf(v - m);
            Sr[j] = e;
            s += e;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float inv = s > 0.f ? 1.f / s : 0.f;
        for (int j = part; j < nk; j += 16) Sr[j] *= inv;
    }
    __syncthreads();
    // ---- C: ctx[i][dd] = sum_j P[i][j] V[dd][j]
    if (w < NC) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* vrow = vb + (size_t)(h * DK + 16 * w + r16) * T;
        for (int kt = 0; kt < nkt; ++kt) {
            const int j = kt * 16 + 4 * g;
            const f32x4 p = *(const f32x4*)(S + r16 * a.SW + j);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j < T) v = *(const f32x4*)(vrow + j);               // T % 4 == 0: the 16-byte piece is inside the row or past it
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p[jj], v[jj], acc, 0, 0, 0);
        }
        const int t0 = i0 + 4 * g;
        if (t0 < T) *(f32x4*)(a.ctx + ((size_t)b * d + h * DK + 16 * w + r16) * T + t0) = acc;
    }
}

static int attn_sw(int T) { return ((T + 15) / 16) * 16 + 4; }
size_t attn_full_lds_bytes(int T) { return (size_t)(16 * attn_sw(T) + 4 * 16 * 33) * sizeof(float); }

hipError_t launch_attn_full(const AttnFull& a, hipStream_t st) {
    if (a.T % 4 || a.T < 1 || (a.DK != 64 && a.DK != 32)) return hipErrorInvalidValue;
    const size_t lds = attn_full_lds_bytes(a.T);
    if (lds > 160 * 1024 || a.T > a.R) return hipErrorInvalidValue;
    AttnFullArgs k{a.qkv, a.ptab, a.bias_u, a.bias_v, a.lens, a.bidir, a.ctx, a.causal, a.H, a.T, a.R, a.PR, attn_sw(a.T)};
    const dim3 grid(ceil_div_h(a.T, 16), a.H, a.B);
    static std::atomic<unsigned long long> d64{0ull}, d32{0ull};
    hipError_t e;
    if (a.DK == 64) {
        if ((e = big_lds(attn_full_kernel<64>, d64)) != hipSuccess) return e;
        hipLaunchKernelGGL(attn_full_kernel<64>, grid, dim3(256), lds, st, k);
    } else {
        if ((e = big_lds(attn_full_kernel<32>, d32)) != hipSuccess) return e;
        hipLaunchKernelGGL(attn_full_kernel<32>, grid, dim3(256), lds, st, k);
    }
    return hipGetLastError();
}

// =====================================================================================================================
// layout changes and input assembly (all HBM-bound elementwise work)
// =====================================================================================================================
__global__ __launch_bounds__(256) void tm_to_fm_kernel(const float* __restrict__ in, const int* __restrict__ lens, int L, int D, int T,
                                                       float* __restrict__ out) {
    // 32 x 32 tile transpose through LDS: read rows of D (coalesced over d), write rows of T (coalesced over t)
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int len = lens ? lens[b] : L;
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, d = d0 + tx;
        tile[r][tx] = (t < len && t < L && d < D) ? in[((size_t)b * L + t) * D + d] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int d = d0 + r, t = t0 + tx;
        if (d < D && t < T) out[((size_t)b * D + d) * T + t] = tile[tx][r];
    }
}
hipError_t launch_tm_to_fm(const float* in, const int* lens, int B, int L, int D, int T, float* out, hipStream_t st) {
    hipLaunchKernelGGL(tm_to_fm_kernel, dim3(ceil_div_h(T, 32), ceil_div_h(D, 32), B), dim3(256), 0, st, in, lens, L, D, T, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void fm_to_tm_kernel(const float* __restrict__ in, const int* __restrict__ lens, int D, int T, int L,
                                                       float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int len = lens ? lens[b] : L;
    for (int r = ty; r < 32; r += 8) {
        const int d = d0 + r, t = t0 + tx;
        tile[r][tx] = (d < D && t < T && t < len) ? in[((size_t)b * D + d) * T + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, d = d0 + tx;
        if (t < L && d < D) out[((size_t)b * L + t) * D + d] = tile[tx][r];
    }
}
hipError_t launch_fm_to_tm(const float* in, const int* lens, int B, int D, int T, int L, float* out, hipStream_t st) {
    hipLaunchKernelGGL(fm_to_tm_kernel, dim3(ceil_div_h(L, 32), ceil_div_h(D, 32), B), dim3(256), 0, st, in, lens, D, T, L, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void token_embed_fm_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table, int vocab,
                                                             int L, int D, int T, float* __restrict__ out) {
    const int b = blockIdx.z, t = blockIdx.x * 64 + (threadIdx.x & 63);
    if (t >= T) return;
    long long id = t < L ? ids[(size_t)b * L + t] : -1;
    if (id >= vocab) id = -1;
    for (int d = blockIdx.y * 4 + (threadIdx.x >> 6); d < D; d += gridDim.y * 4)
        out[((size_t)b * D + d) * T + t] = id >= 0 ? table[(size_t)id * D + d] : 0.f;
}
hipError_t launch_token_embed_fm(const int64_t* ids, const float* table, int vocab, int B, int L, int D, int T, float* out,
                                 hipStream_t st) {
    hipLaunchKernelGGL(token_embed_fm_kernel, dim3(ceil_div_h(T, 64), 8, B), dim3(256), 0, st, ids, table, vocab, L, D, T, out);
    return hipGetLastError();
}

__device__ __forceinline__ float codebook_sum(const float* cb, int K, int D, int nq, const int64_t* tok, int stride, int d) {
    // sum over the groups in group order (QuantizerCodebook.forward, laura_model.py:51-53; cal_codec_emb :303-310)
    float s = 0.f;
    for (int k = 0; k < nq; ++k) {
        long long c = tok[k * stride];
        c = c < 0 ? 0 : (c >= K ? K - 1 : c);
        const float v = cb[((size_t)k * K + c) * D + d];
        s = k == 0 ? v : s + v;
    }
    return s;
}

__global__ __launch_bounds__(256) void lm_assemble_kernel(const float* __restrict__ text_fm, int Tt, const int* __restrict__ text_lens,
                                                          const float* __restrict__ lm_emb, const float* __restrict__ cb, int K, int nq,
                                                          const int64_t* __restrict__ codec, const int* __restrict__ codec_lens, int Cmax,
                                                          int D, int T, float* __restrict__ out, int* seq_len, int* bidir_len) {
    const int b = blockIdx.z, t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int tl = text_lens[b], cl = (codec && codec_lens) ? codec_lens[b] : 0;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        seq_len[b] = tl + 2 + cl;
        if (bidir_len) bidir_len[b] = tl + 1;          // <sos> + text (transformer_lm.py:286-288)
    }
    if (t >= T) return;
    for (int d = blockIdx.y * 4 + (threadIdx.x >> 6); d < D; d += gridDim.y * 4) {
        float v = 0.f;
        if (t == 0) v = lm_emb[d];                                         // sos_eos = 0
        else if (t <= tl) v = text_fm[((size_t)b * D + d) * Tt + (t - 1)];
        else if (t == tl + 1) v = lm_emb[D + d];                           // task_id = 1
        else if (t < tl + 2 + cl) v = codebook_sum(cb, K, D, nq, codec + ((size_t)b * Cmax + (t - tl - 2)) * nq, 1, d);
        out[((size_t)b * D + d) * T + t] = v;
    }
}
hipError_t launch_lm_assemble(const float* text_fm, int Tt, const int* text_lens, const float* lm_emb, const float* cb, int K, int nq,
                              const int64_t* codec, const int* codec_lens, int Cmax, int B, int D, int T, float* out, int* seq_len,
                              int* bidir_len, hipStream_t st) {
    hipLaunchKernelGGL(lm_assemble_kernel, dim3(ceil_div_h(T, 64), 8, B), dim3(256), 0, st, text_fm, Tt, text_lens, lm_emb, cb, K, nq,
                       codec, codec_lens, Cmax, D, T, out, seq_len, bidir_len);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void nar_assemble_kernel(const float* __restrict__ text_fm, int Tt, const int* __restrict__ text_lens,
                                                           const float* __restrict__ cb, int K, int nq, const int64_t* __restrict__ codec,
                                                           const int* __restrict__ codec_lens, int Cmax, int cstride,
                                                           const float* __restrict__ pe_abs, int split, int D, int T,
                                                           float* __restrict__ out, int* seq_len) {
    const int b = blockIdx.z, t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int tl = text_lens[b], cl = codec_lens[b];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) seq_len[b] = tl + cl;
    if (t >= T) return;
    const float xs = sqrtf((float)D);
    for (int d = blockIdx.y * 4 + (threadIdx.x >> 6); d < D; d += gridDim.y * 4) {
        float v = 0.f;
        if (t < tl) {
            v = text_fm[((size_t)b * D + d) * Tt + t];
            if (split) v = v * xs + pe_abs[(size_t)t * D + d];              // PositionalEncoding.forward (embedding.py:79-91)
        } else if (t < tl + cl) {
            v = codebook_sum(cb, K, D, nq, codec + ((size_t)b * Cmax + (t - tl)) * cstride, 1, d);
            if (split) v = v * xs + pe_abs[(size_t)(t - tl) * D + d];
        }
        out[((size_t)b * D + d) * T + t] = v;
    }
}
hipError_t launch_nar_assemble(const float* text_fm, int Tt, const int* text_lens, const float* cb, int K, int nq, const int64_t* codec,
                               const int* codec_lens, int Cmax, int codec_stride_nq, const float* pe_abs, int split, int B, int D, int T,
                               float* out, int* seq_len, hipStream_t st) {
    hipLaunchKernelGGL(nar_assemble_kernel, dim3(ceil_div_h(T, 64), 8, B), dim3(256), 0, st, text_fm, Tt, text_lens, cb, K, nq, codec,
                       codec_lens, Cmax, codec_stride_nq, pe_abs, split, D, T, out, seq_len);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void nar_extract_kernel(const float* __restrict__ in, const int* __restrict__ text_lens,
                                                          const int* __restrict__ codec_lens, int D, int T, int Cmax,
                                                          float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tl = text_lens[b], cl = codec_lens[b];
    for (int r = ty; r < 32; r += 8) {
        const int d = d0 + r, c = c0 + tx, t = tl + c;
        tile[r][tx] = (d < D && c < cl && t < T) ? in[((size_t)b * D + d) * T + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, d = d0 + tx;
        if (c < Cmax && d < D) out[((size_t)b * Cmax + c) * D + d] = tile[tx][r];
    }
}
hipError_t launch_nar_extract(const float* in, const int* text_lens, const int* codec_lens, int B, int D, int T, int Cmax, float* out,
                              hipStream_t st) {
    hipLaunchKernelGGL(nar_extract_kernel, dim3(ceil_div_h(Cmax, 32), ceil_div_h(D, 32), B), dim3(256), 0, st, in, text_lens, codec_lens,
                       D, T, Cmax, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void logsoftmax_fm_kernel(const float* __restrict__ in, const int* __restrict__ lens, int V, int T, int L,
                                                           float* __restrict__ out) {
    const int b = blockIdx.y, t = blockIdx.x * 64 + threadIdx.x;
    if (t >= L) return;
    float* o = out + ((size_t)b * L + t) * V;
    if (t >= lens[b] || t >= T) {
        for (int v = 0; v < V; ++v) o[v] = 0.f;
        return;
    }
    const float* x = in + (size_t)b * V * T + t;
    float m = -INFINITY;
    for (int v = 0; v < V; ++v) m = fmaxf(m, x[(size_t)v * T]);
    float s = 0.f;
    for (int v = 0; v < V; ++v) s += expf(x[(size_t)v * T] - m);
    const float ls = logf(s);
    for (int v = 0; v < V; ++v) o[v] = (x[(size_t)v * T] - m) - ls;
}
hipError_t launch_logsoftmax_fm(const float* in, const int* lens, int B, int V, int T, int L, float* out, hipStream_t st) {
    hipLaunchKernelGGL(logsoftmax_fm_kernel, dim3(ceil_div_h(L, 64), B), dim3(64), 0, st, in, lens, V, T, L, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void kv_store_kernel(const float* __restrict__ qkv, const int* __restrict__ lens, int d, int T, int Tcap,
                                                       float* __restrict__ kc, float* __restrict__ vc) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int len = lens[b];
    const float* kb = qkv + ((size_t)b * 3 + 1) * d * T;
    const float* vb = qkv + ((size_t)b * 3 + 2) * d * T;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, t = t0 + tx;
        const bool ok = n < d && t < len && t < T;
        if (ok) kc[((size_t)b * d + n) * Tcap + t] = kb[(size_t)n * T + t];
        tile[r][tx] = ok ? vb[(size_t)n * T + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, n = n0 + tx;
        if (t < len && t < T && n < d) vc[((size_t)b * Tcap + t) * d + n] = tile[tx][r];
    }
}
hipError_t launch_kv_store(const float* qkv, const int* lens, int B, int d, int T, int Tcap, float* kc, float* vc, hipStream_t st) {
    hipLaunchKernelGGL(kv_store_kernel, dim3(ceil_div_h(T, 32), ceil_div_h(d, 32), B), dim3(256), 0, st, qkv, lens, d, T, Tcap, kc, vc);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void gather_last_kernel(const float* __restrict__ x, const int* __restrict__ lens, int C, int T,
                                                          float* __restrict__ out) {
    const int b = blockIdx.x;
    const int t = lens[b] - 1;
    for (int c = threadIdx.x; c < C; c += 256) out[(size_t)b * C + c] = x[((size_t)b * C + c) * T + (t < 0 ? 0 : t)];
}
hipError_t launch_gather_last(const float* x, const int* lens, int B, int C, int T, float* out, hipStream_t st) {
    hipLaunchKernelGGL(gather_last_kernel, dim3(B), dim3(256), 0, st, x, lens, C, T, out);
    return hipGetLastError();
}

__global__ void fill_i32_kernel(int* p, int v, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) p[i] = v;
}
hipError_t launch_fill_i32(int* p, int v, int n, hipStream_t st) {
    hipLaunchKernelGGL(fill_i32_kernel, dim3(ceil_div_h(n, 64)), dim3(64), 0, st, p, v, n);
    return hipGetLastError();
}

// =====================================================================================================================
// STEP FORM.  Weight-streaming GEMV for B <= 16 new tokens: y[b][n] = f(sum_k W[n][k] g(x)[b][k] + bias[n]).
//   workgroup = one 16-row tile of W, KS waves splitting K; W in fragment order [tile][K/16][lane][4]: every wave load is one
//   contiguous 1 KiB, all loads of a wave's K range issued before its MFMAs; x (optionally LayerNorm'ed) staged in LDS as the
//   B operand (batch = the 16 MFMA columns, rows >= B read a zero row); partial tiles reduced across the waves in wave order
//   (deterministic), then bias / activation / residual add / cache scatter.
// =====================================================================================================================
struct GemvArgs {
    const float *x, *wf, *bias, *gamma, *beta;
    float eps;
    int act, mode;
    float* y;
    int ldy;
    float *kc, *vc;
    const int* pos;
    int d, Tcap, B, K, N, XS;       // XS = LDS row stride of x
    const float* apart;             // attention partials [B][H][NS][DK + 2] (x is then their combination), or null
    int H, DK, NS;
    int ablate;                     // FC_GEMV_ABLATE (tuning aid): 1 no x loads, 2 no weight loads, 4 no LayerNorm, 8 no MFMAs
};

// CPW = 16-wide k chunks per wave as a compile-time constant (2 or 8 for every layer of the recipe): the weight loads are then
// unconditional straight-line code.  Under a run-time count every load sat behind its own `if`, and hipcc waits for a conditional
// load right where it is issued (DESIGN.md §5, round-2 finding) -- the "prefetch" was eight serialised round trips.  CPW = 0: generic.
template <int KS, int CPW>
__global__ __launch_bounds__(64 * KS) void gemv_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                                   // [B + 1][XS], row B = zeros
    float* red = lds + (a.B + 1) * a.XS;               // [KS][64][4]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const int K = a.K, B = a.B, XS = a.XS;
    const int tile = blockIdx.x;
    const int nch = K / 16;                            // 16-wide k chunks of the row
    const int cpw = CPW ? CPW : nch / KS;              // chunks per wave
    // the weight fragments do not depend on x: request them first, the staging / LayerNorm of x runs under their latency
    const float* wp = a.wf + ((size_t)tile * nch + (size_t)w * cpw) * 256 + lane * 4;
    constexpr int NPRE = CPW ? CPW : 1;
    f32x4 wv[CPW ? CPW : 8];
#pragma unroll
    for (int u = 0; u < NPRE; ++u) wv[u] = (a.ablate & 2) ? (f32x4){1.f, 1.f, 1.f, 1.f} : *(const f32x4*)(wp + (size_t)u * 256);
    // epilogue operands (bias, the residual row for mode 1, the cache position for mode 2) requested NOW: fetched at the end of the kernel
    // they are two serial memory round trips with nothing left to overlap them (measured: ~1.5 us of an 8 us kernel)
    const int eb = r16 < B ? r16 : 0, en0 = tile * 16 + 4 * g;
    float ebias[4], eold[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = en0 + r < a.N ? en0 + r : a.N - 1;
        ebias[r] = a.bias[n];
        eold[r] = a.mode == 1 ? a.y[(size_t)eb * a.ldy + n] : 0.f;
    }
    const int epos = a.mode == 2 ? a.pos[eb] : 0;
    if (a.apart) {
        // x[b][h * DK + dd] = combination of the NS key-range partials of head h (flash-decoding): o_s, running max m_s, sum l_s.
        // First the B * H * NS normalised weights exp(m_s - M) / L (one thread per (b, h); red[] is free until the reduction)
        const int PS = a.DK + 2;
        float* wn = red;
        // NS <= 8 key ranges; every loop below runs its 8 trips with clamped indices so that the loads are unconditional and
        // independent (a run-time trip count made each partial its own serialised round trip: 11 us instead of 5 for this kernel)
        for (int e = tid; e < B * a.H; e += 64 * KS) {
            const float* pp = a.apart + (size_t)e * a.NS * PS;
            float mv[8], lv[8];
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) {
                const int sc = sp < a.NS ? sp : a.NS - 1;
                mv[sp] = pp[sc * PS + a.DK];
                lv[sp] = pp[sc * PS + a.DK + 1];
            }
            float M = -INFINITY;
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) M = fmaxf(M, sp < a.NS ? mv[sp] : -INFINITY);
            float L = 0.f, wg[8];
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) {
                wg[sp] = sp < a.NS ? expf(mv[sp] - M) : 0.f;
                L = fmaf(lv[sp], wg[sp], L);
            }
            const float inv = 1.f / L;
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) wn[e * 8 + sp] = wg[sp] * inv;
        }
        __syncthreads();
        for (int e = tid; e < B * K; e += 64 * KS) {
            const int b = e / K, k = e - b * K, h = k / a.DK, dd = k - h * a.DK;
            const float* pp = a.apart + ((size_t)(b * a.H + h) * a.NS) * PS + dd;
            const float* wq = wn + (b * a.H + h) * 8;
            float ov[8];
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) ov[sp] = pp[(sp < a.NS ? sp : a.NS - 1) * PS];
            float o = 0.f;
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) o = fmaf(ov[sp], wq[sp], o);        // weights of ranges >= NS are 0
            Xs[b * XS + k] = o;
        }
        for (int k = tid; k < K; k += 64 * KS) Xs[B * XS + k] = 0.f;
        __syncthreads();                               // wn (= red) is reused by the reduction below
    } else {
        for (int e = tid * 4; e < (B + 1) * K; e += 64 * KS * 4) {
            const int b = e / K, k = e - b * K;
            f32x4 v = (a.ablate & 1) ? (f32x4){1.f, 0.f, 1.f, 0.f} : *(const f32x4*)(a.x + (size_t)(b < B ? b : B - 1) * K + k);      // unconditional load, the zero row by select
            if (b >= B) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            *(f32x4*)(Xs + b * XS + k) = v;
        }
    }
    __syncthreads();
    if (a.gamma && !(a.ablate & 4)) {       // LayerNorm of every row, two-pass in LDS (one wave per row at a time)
        for (int b = w; b < B; b += KS) {
            float* xr = Xs + b * XS;
            float s = 0.f;
            for (int k = lane; k < K; k += 64) s += xr[k];
            const float mean = wave_sum(s) / (float)K;
            float q = 0.f;
            for (int k = lane; k < K; k += 64) { const float dv = xr[k] - mean; q += dv * dv; }
            const float rstd = 1.f / sqrtf(wave_sum(q) / (float)K + a.eps);
            for (int k = lane; k < K; k += 64) xr[k] = (xr[k] - mean) * rstd * a.gamma[k] + a.beta[k];
        }
        __syncthreads();
    }
    const float* xb = Xs + (r16 < B ? r16 : B) * XS + w * cpw * 16 + 4 * g;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const f32x4 xv = *(const f32x4*)(xb + u * 16);
        if (a.ablate & 8) { acc += wv[u] * xv[0]; continue; }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][j], xv[j], acc, 0, 0, 0);
    }
    for (int c0 = NPRE; c0 < cpw; c0 += 8) {           // generic form: the remaining chunks in batches of up to 8
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + u < cpw) wv[u] = *(const f32x4*)(wp + (size_t)(c0 + u) * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + u < cpw) {
                const f32x4 xv = *(const f32x4*)(xb + (c0 + u) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][j], xv[j], acc, 0, 0, 0);
            }
    }
    *(f32x4*)(red + (w * 64 + lane) * 4) = acc;
    __syncthreads();
    if (w != 0) return;
    f32x4 s = *(const f32x4*)(red + lane * 4);
    for (int u = 1; u < KS; ++u) s += *(const f32x4*)(red + (u * 64 + lane) * 4);
    const int b = r16;
    if (b >= B) return;
    const int p = epos;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = tile * 16 + 4 * g + r;
        if (n >= a.N) continue;
        float v = act_f(s[r] + ebias[r], a.act);
        if (a.mode == 0) a.y[(size_t)b * a.ldy + n] = v;
        else if (a.mode == 1) a.y[(size_t)b * a.ldy + n] = eold[r] + v;
        else {
            if (n < a.d) a.y[(size_t)b * a.ldy + n] = v;
            else if (n < 2 * a.d) a.kc[((size_t)b * a.d + (n - a.d)) * a.Tcap + p] = v;
            else a.vc[((size_t)b * a.Tcap + p) * a.d + (n - 2 * a.d)] = v;
        }
    }
}

hipError_t launch_gemv(const Gemv& g, hipStream_t st) {
    if (g.B < 1 || g.B > 16 || g.K % 16 || g.K < 16) return hipErrorInvalidValue;
    const int nch = g.K / 16;
    static const int ks_cap = fc::ab_knob("FC_GEMV_KS", 16);     // tuning aid: waves per workgroup
    int KS = ks_cap >= 1 && ks_cap <= 16 ? ks_cap : 16;
    while (KS > 1 && (nch % KS || nch / KS < 2)) KS >>= 1;     // >= 2 chunks per wave, K split evenly
    if (nch % KS) KS = 1;
    GemvArgs a{g.x, g.wf, g.bias, g.gamma, g.beta, g.eps, g.act, g.mode, g.y, g.ldy, g.kc, g.vc, g.pos, g.d, g.Tcap, g.B, g.K, g.N, g.K + 4,
               g.apart, g.H, g.DK, g.NS, 0};
    static const int ablate = fc::ab_knob("FC_GEMV_ABLATE", 0);
    a.ablate = ablate;
    if (g.apart && (g.H * g.DK != g.K || g.NS < 1 || g.NS > 8 || g.B * g.H * 8 > KS * 256)) return hipErrorInvalidValue;
    const size_t lds = ((size_t)(g.B + 1) * a.XS + (size_t)KS * 256) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid(ceil_div_h(g.N, 16));
    static std::atomic<unsigned long long> d16[6], d8[6], d4[6], d2[6], d1[6];
    hipError_t e;
#define FC_GEMV_CASE(ks, flag)                                                            \
    if (KS == ks) {                                                                       \
        const int cpw = nch / ks;                                                         \
        if (cpw == 2) {                                                                   \
            if ((e = big_lds(gemv_kernel<ks, 2>, flag[0])) != hipSuccess) return e;       \
            hipLaunchKernelGGL((gemv_kernel<ks, 2>), grid, dim3(64 * ks), lds, st, a);    \
        } else if (cpw == 8) {                                                            \
            if ((e = big_lds(gemv_kernel<ks, 8>, flag[1])) != hipSuccess) return e;       \
            hipLaunchKernelGGL((gemv_kernel<ks, 8>), grid, dim3(64 * ks), lds, st, a);    \
        } else if (cpw == 4) {                                                            \
            if ((e = big_lds(gemv_kernel<ks, 4>, flag[3])) != hipSuccess) return e;       \
            hipLaunchKernelGGL((gemv_kernel<ks, 4>), grid, dim3(64 * ks), lds, st, a);    \
        } else if (cpw == 16) {                                                           \
            if ((e = big_lds(gemv_kernel<ks, 16>, flag[4])) != hipSuccess) return e;      \
            hipLaunchKernelGGL((gemv_kernel<ks, 16>), grid, dim3(64 * ks), lds, st, a);   \
        } else if (cpw == 32) {                                                           \
            if ((e = big_lds(gemv_kernel<ks, 32>, flag[5])) != hipSuccess) return e;      \
            hipLaunchKernelGGL((gemv_kernel<ks, 32>), grid, dim3(64 * ks), lds, st, a);   \
        } else {                                                                          \
            if ((e = big_lds(gemv_kernel<ks, 0>, flag[2])) != hipSuccess) return e;       \
            hipLaunchKernelGGL((gemv_kernel<ks, 0>), grid, dim3(64 * ks), lds, st, a);    \
        }                                                                                 \
        return hipGetLastError();                                                         \
    }
    FC_GEMV_CASE(16, d16)
    FC_GEMV_CASE(8, d8)
    FC_GEMV_CASE(4, d4)
    FC_GEMV_CASE(2, d2)
    FC_GEMV_CASE(1, d1)
#undef FC_GEMV_CASE
    return hipErrorInvalidValue;
}

__global__ __launch_bounds__(64) void layernorm_rows_kernel(float* x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, int relu, float post_scale, int C) {
    float* xr = x + (size_t)blockIdx.x * C;
    const int lane = threadIdx.x;
    float s = 0.f;
    for (int k = lane; k < C; k += 64) s += xr[k];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int k = lane; k < C; k += 64) { const float dv = xr[k] - mean; q += dv * dv; }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
    for (int k = lane; k < C; k += 64) {
        float o = (xr[k] - mean) * rstd * gamma[k] + beta[k];
        if (relu) o = o > 0.f ? o : 0.f;
        xr[k] = o * post_scale;
    }
}
hipError_t launch_layernorm_rows(float* x, const float* gamma, const float* beta, float eps, int relu, float post_scale, int B, int C,
                                 hipStream_t st) {
    hipLaunchKernelGGL(layernorm_rows_kernel, dim3(B), dim3(64), 0, st, x, gamma, beta, eps, relu, post_scale, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Step attention: one query (the newest token) of one (utterance, head) against the KV cache.  K cache feature-major
// [dd][j] (lanes over keys: coalesced), V cache token-major [j][dd] (lanes over dims: coalesced); relative position of key j
// is pos - j >= 0, read backwards from the position table.
// ---------------------------------------------------------------------------------------------------------------------
struct AttnStepArgs {
    const float *q, *kc, *vc, *ptab, *bias_u, *bias_v;
    const int* pos;
    float* part;         // [B][H][NS][DK + 2]: o[DK] (unnormalised), running max, sum of exponentials
    int H, Tcap, R, PR, NS;
};

// One query (the newest token) of one (utterance, head) against ONE key range of the KV cache (flash-decoding split: NS ranges,
// combined by the consumer, gemv_kernel's `apart` prologue).  The kernel is latency-bound, so every load of a thread is an
// independent 16-byte piece and the K / position-table / V loads of the first pass are all in flight together:
//   scores: thread (dg, q) owns DK/8 dims of 4 consecutive keys (K cache rows are key-contiguous; the position table is read
//           backwards: key j sits at column R-1+pos-j); the 8 dim-group partials meet in LDS;
//   context: thread (jg, dq) owns 4 dims of every 16th key (V cache rows are dim-contiguous).
template <int DK>
__global__ __launch_bounds__(256) void attn_step_kernel(AttnStepArgs a) {
    constexpr int DG = 8, DPG = DK / DG;             // dim groups, dims per group
    constexpr int NQ = DK / 4, NG = 256 / NQ;        // dim quads, key groups of the context pass
    constexpr int CH = 128;                          // keys per pass
    __shared__ __attribute__((aligned(16))) float part[DG][CH];
    __shared__ float sc[CH];
    __shared__ float qu[DK], qv[DK];
    __shared__ float wred[4];
    __shared__ __attribute__((aligned(16))) float cred[NG][DK];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y, sp = blockIdx.z, d = a.H * DK;
    const int p = a.pos[b], n = p + 1;
    int chunk = ((n + a.NS - 1) / a.NS + 3) & ~3;    // keys per split, whole quads
    const int k0 = sp * chunk;
    const int k1 = k0 + chunk < n ? k0 + chunk : n;
    float* out = a.part + ((size_t)(b * a.H + h) * a.NS + sp) * (DK + 2);
    if (k0 >= k1) {                                  // empty range: weight 0 in the combination
        if (tid < DK) out[tid] = 0.f;
        if (tid == 0) { out[DK] = -INFINITY; out[DK + 1] = 0.f; }
        return;
    }
    if (tid < DK) {
        const float q = a.q[(size_t)b * d + h * DK + tid];
        qu[tid] = q + a.bias_u[h * DK + tid];
        qv[tid] = q + a.bias_v[h * DK + tid];
    }
    const int dg = tid >> 5, ql = tid & 31;          // scores: dim group, quad of the pass
    const int dq = tid % NQ, jg = tid / NQ;          // context: dim quad, key group
    const float* kb = a.kc + ((size_t)b * d + h * DK + dg * DPG) * a.Tcap;
    const float* pb = a.ptab + (size_t)(h * DK + dg * DPG) * a.PR + (a.R - 1) + p;
    const float* vb = a.vc + (size_t)b * a.Tcap * d + h * DK + 4 * dq;
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o_acc = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = k0; c0 < k1; c0 += CH) {
        const int cn = k1 - c0 < CH ? k1 - c0 : CH;  // keys of this pass
        // ---- all loads of the pass: K / position rows for the scores, V rows for the context
        const int j0 = c0 + 4 * ql;
        const bool sq = 4 * ql < cn;
        // straight-line loads with clamped addresses: hipcc waits for a load under a run-time condition right where it is issued, which
        // would serialise the round trips this kernel exists to overlap.  Quads / rows past the range read key c0 and are never used.
        const int jq = sq ? j0 : c0;
        f32x4 kv[DPG], pv[DPG];
#pragma unroll
        for (int dd = 0; dd < DPG; ++dd) {
            kv[dd] = *(const f32x4*)(kb + (size_t)dd * a.Tcap + jq);
            pv[dd] = *(const f32x4u*)(pb + (size_t)dd * a.PR - jq - 3);     // columns of keys jq+3, jq+2, jq+1, jq
        }
        constexpr int NV = CH / NG;                  // V rows per thread and pass
        f32x4 vv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int jj = jg + i * NG;
            vv[i] = *(const f32x4*)(vb + (size_t)(c0 + (jj < cn ? jj : 0)) * d);
        }
        __syncthreads();                             // qu / qv visible (first pass); LDS of the previous pass free
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dd = 0; dd < DPG; ++dd) {
            const float u = qu[dg * DPG + dd], v = qv[dg * DPG + dd];
            acc[0] = fmaf(u, kv[dd][0], fmaf(v, pv[dd][3], acc[0]));
            acc[1] = fmaf(u, kv[dd][1], fmaf(v, pv[dd][2], acc[1]));
            acc[2] = fmaf(u, kv[dd][2], fmaf(v, pv[dd][1], acc[2]));
            acc[3] = fmaf(u, kv[dd][3], fmaf(v, pv[dd][0], acc[3]));
        }
        *(f32x4*)&part[dg][4 * ql] = acc;
        __syncthreads();
        const float scale = 1.f / sqrtf((float)DK);
        float s = -INFINITY;
        if (tid < cn) {
            float t = 0.f;
#pragma unroll
            for (int gq = 0; gq < DG; ++gq) t += part[gq][tid];
            s = t * scale;
        }
        float m = wave_max(s);
        if (lane == 0) wred[w] = m;
        __syncthreads();
        m = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
        const float m_new = fmaxf(m_run, m);
        const float e = tid < cn ? expf(s - m_new) : 0.f;
        if (tid < CH) sc[tid] = e;
        float l = wave_sum(e);
        __syncthreads();                             // wred read by all; sc written
        if (lane == 0) wred[w] = l;
        __syncthreads();
        const float corr = expf(m_run - m_new);      // 0 on the first pass (m_run = -inf)
        l_run = l_run * corr + (wred[0] + wred[1] + wred[2] + wred[3]);
        m_run = m_new;
        o_acc *= corr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int jj = jg + i * NG;
            if (jj < cn) o_acc += sc[jj] * vv[i];
        }
    }
    *(f32x4*)&cred[jg][4 * dq] = o_acc;
    __syncthreads();
    if (tid < DK) {
        float o = 0.f;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) o += cred[gq][tid];
        out[tid] = o;
    }
    if (tid == 0) { out[DK] = m_run; out[DK + 1] = l_run; }
}

hipError_t launch_attn_step(const AttnStep& a, hipStream_t st) {
    if ((a.DK != 64 && a.DK != 32) || a.NS < 1 || a.Tcap % 4) return hipErrorInvalidValue;
    AttnStepArgs k{a.q, a.kc, a.vc, a.ptab, a.bias_u, a.bias_v, a.pos, a.part, a.H, a.Tcap, a.R, a.PR, a.NS};
    if (a.DK == 64) hipLaunchKernelGGL(attn_step_kernel<64>, dim3(a.H, a.B, a.NS), dim3(256), 0, st, k);
    else hipLaunchKernelGGL(attn_step_kernel<32>, dim3(a.H, a.B, a.NS), dim3(256), 0, st, k);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Device-side token sampling (LauraGenModel.sampling_ids, laura_model.py:466-499; the loop body of decode_codec :537-542).
// One workgroup per utterance, the nq groups in turn.  The reference samples from softmax(log_softmax(logits)[group]) =
// softmax(logits[group]).  The multinomial draw is the inverse CDF of ONE uniform number per (utterance, step, group) from a
// counter-based generator (Philox4x32-10 keyed by the caller's seed), so a generation is reproducible from its seed and never
// leaves the device; torch.multinomial's own generator stream cannot be shared by any second implementation.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mulhi32(unsigned a, unsigned b) { return __umulhi(a, b); }
__device__ float philox_uniform(unsigned long long seed, unsigned c0, unsigned c1, unsigned c2) {
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    unsigned x0 = c0, x1 = c1, x2 = c2, x3 = 0x5eedu;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned h0 = mulhi32(0xD2511F53u, x0), l0 = 0xD2511F53u * x0;
        const unsigned h1 = mulhi32(0xCD9E8D57u, x2), l1 = 0xCD9E8D57u * x2;
        const unsigned n0 = h1 ^ x1 ^ k0, n1 = l1, n2 = h0 ^ x3 ^ k1, n3 = l0;
        x0 = n0; x1 = n1; x2 = n2; x3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(x0 >> 8) * (1.0f / 16777216.0f);      // [0, 1)
}

struct SampleArgs {
    const float* logits;
    int K, nq, mode, ki;
    float pf;
    unsigned long long seed;
    const int64_t* forced;
    int max_steps;
    int64_t* tokens;
    int tok_stride;
    const int* tok_off;
    int *n_gen, *done, *n_done, *pos, *step;
    float* logp_out;
    const float* cb;
    int D;
    float* next_emb;
    int B;
    // fused input layer of the LM for the next step (TransformerEncoder_s0.embed, transformer_encoder.py:463-470):
    // xs[b] = [relu](LayerNorm(W e + bias)) * xscale;  emb_wt [D][dm] (W transposed), null = not fused
    const float *emb_wt, *emb_bias, *emb_g, *emb_b;
    int dm, emb_relu;
    float xscale;
    float* xs;
    unsigned* launch_seq;
};

#define FC_SAMPLE_MAXV 2048      // candidates per group, padded to a power of two for the bitonic sort (K + 1 <= 2048)

__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
    __shared__ __attribute__((aligned(16))) float val[FC_SAMPLE_MAXV];
    __shared__ __attribute__((aligned(16))) int idx[FC_SAMPLE_MAXV];
    __shared__ float csum[256];
    __shared__ float wred[4];
    __shared__ int wredi[4];
    __shared__ int chosen[8];
    __shared__ float sh_f[2];
    __shared__ int sh_i[2];
    __shared__ unsigned hist[256];
    __shared__ float cval[256];
    __shared__ int cidx[256];
    __shared__ int ccount;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x;
    const int G = a.K + 1, V = a.nq * G;
    const int step = a.step[b];          // per-utterance sample counter (RNG stream position)
    const int gen = a.n_gen[b];
    const bool was_done = a.done[b] != 0;
    const float* lg = a.logits + (size_t)b * V;
    // log-softmax over the whole vocabulary of the step (TransformerEmbedLM.score, transformer_lm.py:309-311), when asked for
    if (a.logp_out && !was_done && gen < a.max_steps) {
        float m = -INFINITY;
        for (int v = tid; v < V; v += 256) m = fmaxf(m, lg[v]);
        m = wave_max(m);
        if (lane == 0) wred[w] = m;
        __syncthreads();
        m = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
        __syncthreads();
        float s = 0.f;
        for (int v = tid; v < V; v += 256) s += expf(lg[v] - m);
        s = wave_sum(s);
        if (lane == 0) wred[w] = s;
        __syncthreads();
        const float ls = logf(wred[0] + wred[1] + wred[2] + wred[3]);
        float* o = a.logp_out + ((size_t)b * a.max_steps + gen) * V;
        for (int v = tid; v < V; v += 256) o[v] = (lg[v] - m) - ls;
        __syncthreads();
    }
    for (int k = 0; k < a.nq; ++k) {
        const float* x = lg + k * G;
        // ---- group maximum and its FIRST index (greedy: weighted_scores.topk(1))
        // the group's logits, once, as 8 unconditional clamped loads per thread (a `for (v = tid; v < G; v += 256)` loop waits for
        // every load before the next: three such passes per group were most of this kernel's 33 us)
        constexpr int XPT = FC_SAMPLE_MAXV / 256;
        float xr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) { const int v = tid + 256 * i; xr[i] = x[v < G ? v : G - 1]; }
        float m = -INFINITY;
        int mi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int v = tid + 256 * i;
            if (v < G && xr[i] > m) { m = xr[i]; mi = v; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o, 64);
            const int oi = __shfl_xor(mi, o, 64);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        if (lane == 0) { wred[w] = m; wredi[w] = mi; }
        __syncthreads();
        m = wred[0]; mi = wredi[0];
        for (int u = 1; u < 4; ++u)
            if (wred[u] > m || (wred[u] == m && wredi[u] < mi)) { m = wred[u]; mi = wredi[u]; }
        __syncthreads();
        int pick = mi;
        if (a.mode != 0) {
            // unnormalised probabilities exp(x - max) (the normaliser cancels in every mode except the nucleus threshold)
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int v = tid + 256 * i;
                const float e = v < G ? expf(xr[i] - m) : -1.f;      // padding sorts last
                val[v] = e;
                idx[v] = v;
                if (v < G) s += e;
            }
            s = wave_sum(s);
            if (lane == 0) wred[w] = s;
            __syncthreads();
            const float total = wred[0] + wred[1] + wred[2] + wred[3];
            int ncand = G;
            bool selected = false;                   // top-k candidates already in val[] / idx[] order
            if (a.mode == 2 && a.ki <= 64) {
                // top-k with a small k (the recipe samples with --sampling 25): radix-select the k-th largest logit (4 passes over
                // 8-bit digits of the order-preserving integer image of the float), collect everything >= it, order those few by
                // (probability descending, index ascending) with a counting rank -- no full sort of the 1025 logits
                const int kk = a.ki < 1 ? 1 : a.ki;
                unsigned lk[XPT];
                int nl = 0;
#pragma unroll
                for (int i = 0; i < XPT; ++i) {
                    const unsigned u = __float_as_uint(xr[i]);
                    lk[i] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                    if (tid + 256 * i < G) nl = i + 1;
                }
                unsigned prefix = 0u, mask = 0u;
                int need = kk;
                for (int shift = 24; shift >= 0; shift -= 8) {
                    hist[tid] = 0u;
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < XPT; ++i)
                        if (i < nl && (lk[i] & mask) == prefix) atomicAdd(&hist[(lk[i] >> shift) & 255u], 1u);
                    __syncthreads();
                    if (w == 0) {        // wave 0: lane l owns bins 4l .. 4l+3; suffix sums (bins above) by shuffles, then one lane walks its 4 bins
                        const int h0 = (int)hist[4 * lane], h1 = (int)hist[4 * lane + 1], h2 = (int)hist[4 * lane + 2], h3 = (int)hist[4 * lane + 3];
                        const int mine = h0 + h1 + h2 + h3;
                        int incl = mine;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) {
                            const int t = __shfl_down(incl, o, 64);
                            if (lane + o < 64) incl += t;
                        }
                        const int above = incl - mine;               // entries in bins of higher lanes
                        if (above < need && need <= incl) {
                            int cum = above, bin = 4 * lane + 3;
                            const int hh[4] = {h0, h1, h2, h3};
                            for (; bin > 4 * lane; --bin) {
                                if (cum + hh[bin & 3] >= need) break;
                                cum += hh[bin & 3];
                            }
                            sh_i[0] = bin;
                            sh_i[1] = need - cum;
                        }
                    }
                    __syncthreads();
                    prefix |= (unsigned)sh_i[0] << shift;
                    need = sh_i[1];
                    mask |= 255u << shift;
                    __syncthreads();
                }
                if (tid == 0) ccount = 0;
                __syncthreads();
#pragma unroll
                for (int i = 0; i < XPT; ++i) {
                    const int v = tid + 256 * i;
                    if (i < nl && lk[i] >= prefix) {
                        const int slot = atomicAdd(&ccount, 1);
                        if (slot < 256) { cval[slot] = val[v]; cidx[slot] = v; }
                    }
                }
                __syncthreads();
                // more than 256 logits at or above the k-th key (hundreds of ties at the threshold: saturated or constant logits): which of
                // them the 256 slots hold depends on the order of the atomics -- take the full sort below instead, whose tie-break
                // (probability descending, index ascending) is the documented one.  ccount is uniform here.
                if (ccount <= 256) {
                    const int nc = ccount;
                    float myv = 0.f;
                    int myi = 0, rank = 0;
                    if (tid < nc) {
                        myv = cval[tid]; myi = cidx[tid];
                        for (int c = 0; c < nc; ++c) {
                            const float ov = cval[c];
                            const int oi = cidx[c];
                            rank += (ov > myv || (ov == myv && oi < myi)) ? 1 : 0;
                        }
                    }
                    __syncthreads();
                    if (tid < nc) { val[rank] = myv; idx[rank] = myi; }
                    __syncthreads();
                    ncand = kk < nc ? kk : nc;
                    selected = true;
                }
            }
            if (!selected && a.mode >= 2) {
                // descending by probability, ties by index (topk / a stable descending sort): bitonic sort of 2048 pairs
                for (int kk = 2; kk <= FC_SAMPLE_MAXV; kk <<= 1)
                    for (int j = kk >> 1; j > 0; j >>= 1) {
                        __syncthreads();
                        for (int i = tid; i < FC_SAMPLE_MAXV; i += 256) {
                            const int l = i ^ j;
                            if (l > i) {
                                const bool up = (i & kk) == 0;      // "up" = descending order here
                                const float vi = val[i], vl = val[l];
                                const int ii = idx[i], il = idx[l];
                                const bool i_first = vi > vl || (vi == vl && ii < il);
                                if (up ? !i_first : i_first) { val[i] = vl; val[l] = vi; idx[i] = il; idx[l] = ii; }
                            }
                        }
                    }
                __syncthreads();
                if (a.mode == 2) ncand = a.ki < G ? (a.ki < 1 ? 1 : a.ki) : G;
                else {
                    // nucleus: take entries while the running probability mass is < pf (the entry that crosses is kept)
                    if (tid == 0) {
                        float cum = 0.f;
                        int n = 0;
                        const float thr = a.pf * total;
                        while (n < G && cum < thr) { cum += val[n]; ++n; }
                        sh_i[0] = n < 1 ? 1 : n;
                    }
                    __syncthreads();
                    ncand = sh_i[0];
                }
            }
            // inverse CDF over the first ncand candidates in their order: chunk sums, then an exact walk inside the chunk
            const int CH = (ncand + 255) / 256;
            float cs = 0.f;
            for (int i = tid * CH; i < (tid + 1) * CH && i < ncand; ++i) cs += val[i];
            csum[tid] = cs;
            __syncthreads();
            if (w == 0) {        // wave 0: lane l owns chunks 4l .. 4l+3; prefix sums by shuffles, then one lane walks its chunks / elements
                const float c0 = csum[4 * lane], c1 = csum[4 * lane + 1], c2 = csum[4 * lane + 2], c3 = csum[4 * lane + 3];
                const float mine = (c0 + c1) + (c2 + c3);
                float incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const float t = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t;
                }
                const float tot = __shfl(incl, 63, 64);
                const float u = philox_uniform(a.seed, (unsigned)step, (unsigned)b, (unsigned)k);
                const float target = u * tot;
                const float below = incl - mine;
                // the lane whose range [below, incl) holds the target; rounding at the very top goes to the last non-empty lane
                const unsigned long long nonempty = __ballot(mine > 0.f);
                const unsigned long long ballot = __ballot(mine > 0.f && target >= below && target < incl);
                const int owner = ballot ? (int)__ffsll((long long)ballot) - 1 : (nonempty ? 63 - __clzll((long long)nonempty) : 0);
                if (lane == owner) {
                    const float cc[4] = {c0, c1, c2, c3};
                    float run = below;
                    int c = 0;
                    while (c < 3 && run + cc[c] <= target) { run += cc[c]; ++c; }
                    int i = (4 * lane + c) * CH;
                    const int iend = (4 * lane + c + 1) * CH < ncand ? (4 * lane + c + 1) * CH : ncand;
                    while (i + 1 < iend && run + val[i] <= target) { run += val[i]; ++i; }
                    if (i >= ncand) i = ncand - 1;
                    if (i < 0) i = 0;
                    sh_i[1] = idx[i];
                }
            }
            __syncthreads();
            pick = sh_i[1];
            __syncthreads();
        }
        if (tid == 0) {
            if (a.forced && gen < a.max_steps) pick = (int)a.forced[((size_t)b * a.max_steps + gen) * a.nq + k];
            chosen[k] = pick;
        }
        __syncthreads();
    }
    // ---- bookkeeping of decode_codec (laura_model.py:519-546): a step whose ids contain <eos> ends the utterance and is dropped
    bool eos = false;
    for (int k = 0; k < a.nq; ++k) eos = eos || chosen[k] == a.K;
    const bool active = !was_done && gen < a.max_steps;
    if (active && !eos) {
        if (tid < a.nq) a.tokens[((size_t)b * a.tok_stride + a.tok_off[b] + gen) * a.nq + tid] = (int64_t)chosen[tid];
        // next LM input: sum of the groups' codebook rows (build_llm_io -> calc_dense_vector)
        for (int dd = tid; dd < a.D; dd += 256) {
            float s = 0.f;
            for (int k = 0; k < a.nq; ++k) {
                const float v = a.cb[((size_t)k * a.K + chosen[k]) * a.D + dd];
                s = k == 0 ? v : s + v;
            }
            a.next_emb[(size_t)b * a.D + dd] = s;
            val[dd] = s;                                   // val[] is free now: the embedding for the fused input layer
        }
        if (a.emb_wt) {
            __syncthreads();
            // y[n] = bias[n] + sum_k W^T[k][n] e[k]: thread (output quad nq4, k part kp) takes 4 consecutive outputs over its share of
            // the k range with independent 16-byte loads; the parts meet in LDS (idx[] reinterpreted, free now)
            float* ysum = (float*)idx;                      // [dm] after the reduction
            const int nquads = a.dm >> 2;                   // dm % 4 == 0
            const int kparts = 256 / nquads > 0 ? 256 / nquads : 1;
            const int nq4 = tid % nquads, kp = tid / nquads;
            const int kper = (a.D + kparts - 1) / kparts;
            f32x4 accq = {0.f, 0.f, 0.f, 0.f};
            if (kp < kparts && tid < nquads * kparts) {
                const int kb0 = kp * kper, kb1 = kb0 + kper < a.D ? kb0 + kper : a.D;
                const float* wq = a.emb_wt + 4 * nq4;
                int kk = kb0;
                for (; kk + 16 <= kb1; kk += 16) {          // 16 independent 16-byte loads in flight (the table is cold in L2 every step)
                    f32x4 wv16[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) wv16[u] = *(const f32x4*)(wq + (size_t)(kk + u) * a.dm);
#pragma unroll
                    for (int u = 0; u < 16; ++u) accq += val[kk + u] * wv16[u];
                }
                for (; kk + 4 <= kb1; kk += 4) {
                    f32x4 wv4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) wv4[u] = *(const f32x4*)(wq + (size_t)(kk + u) * a.dm);
#pragma unroll
                    for (int u = 0; u < 4; ++u) accq += val[kk + u] * wv4[u];
                }
                for (; kk < kb1; ++kk) accq += val[kk] * *(const f32x4*)(wq + (size_t)kk * a.dm);
            }
            __syncthreads();                                // everybody is done with idx[] (the sampler's candidate ids)
            float* ypart = (float*)idx;                     // [kparts][dm], kparts * dm <= 2048 floats
            if (tid < nquads * kparts) *(f32x4*)(ypart + kp * a.dm + 4 * nq4) = accq;
            __syncthreads();
            float yv[4];
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = tid + 256 * i;
                yv[i] = 0.f;
                if (n < a.dm) {
                    float acc = a.emb_bias[n];
                    for (int q = 0; q < kparts; ++q) acc += ypart[q * a.dm + n];
                    yv[i] = acc;
                    ps += acc;
                }
            }
            (void)ysum;
            ps = wave_sum(ps);
            if (lane == 0) wred[w] = ps;
            __syncthreads();
            const float mean = (wred[0] + wred[1] + wred[2] + wred[3]) / (float)a.dm;
            __syncthreads();
            float pq = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tid + 256 * i < a.dm) { const float dv = yv[i] - mean; pq += dv * dv; }
            pq = wave_sum(pq);
            if (lane == 0) wred[w] = pq;
            __syncthreads();
            const float rstd = 1.f / sqrtf((wred[0] + wred[1] + wred[2] + wred[3]) / (float)a.dm + 1e-5f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = tid + 256 * i;
                if (n < a.dm) {
                    float o = (yv[i] - mean) * rstd * a.emb_g[n] + a.emb_b[n];
                    if (a.emb_relu) o = o > 0.f ? o : 0.f;
                    a.xs[(size_t)b * a.dm + n] = o * a.xscale;
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (active) {
            if (eos) { a.done[b] = 1; atomicAdd(a.n_done, 1); }
            else {
                a.n_gen[b] = gen + 1;
                if (step > 0) a.pos[b] += 1;      // step 0 samples from the prefix; later steps appended one token to the cache
                if (gen + 1 >= a.max_steps) { a.done[b] = 1; atomicAdd(a.n_done, 1); }
            }
        }
        a.step[b] = step + 1;
        if (b == 0 && a.launch_seq) *a.launch_seq += 1u;
    }
}

hipError_t launch_sample(const Sample& s, hipStream_t st) {
    if (s.K + 1 > FC_SAMPLE_MAXV || s.nq > 8 || s.nq < 1) return hipErrorInvalidValue;
    if (s.emb_wt && (s.dm > 1024 || s.dm % 4 || s.D > FC_SAMPLE_MAXV)) return hipErrorInvalidValue;
    SampleArgs a{s.logits, s.K, s.nq, s.mode, s.ki, s.pf, s.seed, s.forced, s.max_steps, s.tokens, s.tok_stride, s.tok_off, s.n_gen,
                 s.done, s.n_done, s.pos, s.step, s.logp_out, s.cb, s.D, s.next_emb, s.B,
                 s.emb_wt, s.emb_bias, s.emb_g, s.emb_b, s.dm, s.emb_relu, s.xscale, s.xs, s.launch_seq};
    hipLaunchKernelGGL(sample_kernel, dim3(s.B), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace laura
}  // namespace fc
