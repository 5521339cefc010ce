/*
 * TEST INFRASTRUCTURE ONLY (oracle).  Never linked into or called by the product path.
 *
 * Plain-C restatement of the residual vector quantiser of the reference:
 *   DistributedResidualVectorQuantization.forward   funcodec/modules/quantization/ddp_core_vq.py:367-418 (eval)
 *   EuclideanCodebook.quantize                       ddp_core_vq.py:180-188
 *   EuclideanCodebook.dequantize                     ddp_core_vq.py:190-192
 *   DistributedResidualVectorQuantization.decode     ddp_core_vq.py:442-453
 *
 *   stage i:  dist[k] = -(( |x|^2 - (2x).e_k ) + |e_k|^2),  idx = argmax_k (first maximum wins, like
 *             torch.max(dim).indices on CPU), q = E_i[idx], residual -= q, out += q.
 *
 * The reference evaluates the three terms with ATen kernels whose summation order is unspecified (MKL sgemm,
 * vectorised reductions).  This file fixes ONE explicit fp32 order -- the one the gfx950 kernel uses
 * (funcodec_amd/csrc/kernels.hip, rvq_encode_kernel) -- so that the integer output can be compared
 * BIT-EXACTLY between CPU and GPU at any size, while tests/golden/rvq_*.npz (produced by the real
 * reference) pin that this order reproduces the reference's indices:
 *   |x|^2  : four partial chains over d in [j*D/4,(j+1)*D/4), s = s + fl(x*x) (square rounded separately,
 *            as x.pow(2).sum(1) does), combined (p0+p1)+(p2+p3);
 *   (2x).e : fmaf chain from 0 in the order d = 16q + 4g + j for q = 0..D/16-1, j = 0..3, g = 0..3 (g innermost),
 *            which is the k order of v_mfma_f32_16x16x4_f32 over the kernel's operand layout;
 *   |e|^2  : sequential d = 0..D-1, squares rounded separately (embed.pow(2).sum(0)).
 *
 * Build: make -C oracle/c   (gcc -O2 -ffp-contract=off; fmaf() calls are explicit)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float sq_norm_seq(const float* e, int D) {
    volatile float s = 0.f;
    for (int d = 0; d < D; ++d) {
        volatile float sq = e[d] * e[d];
        s = s + sq;
    }
    return s;
}

static float sq_norm_4chains(const float* x, int D) {
    float p[4];
    for (int j = 0; j < 4; ++j) {
        volatile float s = 0.f;
        for (int d = j * (D / 4); d < (j + 1) * (D / 4); ++d) {
            volatile float sq = x[d] * x[d];
            s = s + sq;
        }
        p[j] = s;
    }
    volatile float a = p[0] + p[1];
    volatile float b = p[2] + p[3];
    volatile float r = a + b;
    return r;
}

/* (2x).e in the MFMA k order */
static float dot2_mfma_order(const float* x, const float* e, int D) {
    float acc = 0.f;
    for (int q = 0; q < D / 16; ++q)
        for (int j = 0; j < 4; ++j)
            for (int g = 0; g < 4; ++g) {
                const int d = 16 * q + 4 * g + j;
                acc = fmaf(x[d] + x[d], e[d], acc);
            }
    return acc;
}

/* x [N][D] rows, cb [nq][K][D]; codes [nq][N] (int64), quant [N][D] (may be NULL).
 * Returns 0, or -1 on bad arguments. */
int rvq_oracle_encode_src0(const float* x, int N, int D, int K, int nq, const float* cb, int64_t* codes, float* quant, const int32_t* src0);
int rvq_oracle_encode(const float* x, int N, int D, int K, int nq, const float* cb, int64_t* codes, float* quant) {
    return rvq_oracle_encode_src0(x, N, D, K, nq, cb, codes, quant, NULL);
}

/* torch's nearest-neighbour source index for F.interpolate(mode="nearest") on CPU (ATen/native/UpSample.h nearest_idx,
 * cpu/UpSampleKernel.cpp HelperInterpNearest): scale = float(in) / out in fp32, src = min(int(floorf(dst * scale)), in - 1). */
static int nearest_src(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    int src = (int)floorf((float)dst * scale);
    return src < in - 1 ? src : in - 1;
}

/* quantizer_conf.q0_ds_ratio > 1, ddp_core_vq.py:396-404: stage 0 runs on F.interpolate(residual, size=[Tf // 2]) and its
 * quantised output and indices return through F.interpolate(size=[Tf]).  Stage 0 is row-wise, so frame t ends up with the
 * stage-0 result of frame src[t] = down[up[t]], where down[j] = nearest_src(j, Tf, Tf / 2) and up[t] = nearest_src(t, Tf / 2, Tf). */
int rvq_oracle_q0_source(int Tf, int32_t* src) {
    const int half = Tf / 2;
    if (half < 1) return -1;
    for (int t = 0; t < Tf; ++t) src[t] = nearest_src(nearest_src(t, half, Tf), Tf, half);
    return 0;
}

/* src0 != NULL: stage 0 of row n quantises row src0[n] (q0_ds_ratio, above); every later stage and the residual are row n's own */
int rvq_oracle_encode_src0(const float* x, int N, int D, int K, int nq, const float* cb, int64_t* codes, float* quant, const int32_t* src0) {
    if (D % 16 != 0 || N < 0) return -1;
    float* enorm = (float*)malloc(sizeof(float) * (size_t)nq * K);
    float* res = (float*)malloc(sizeof(float) * D);
    float* out = (float*)malloc(sizeof(float) * D);
    for (size_t r = 0; r < (size_t)nq * K; ++r) enorm[r] = sq_norm_seq(cb + r * D, D);
    for (int n = 0; n < N; ++n) {
        memcpy(res, x + (size_t)n * D, sizeof(float) * D);
        for (int d = 0; d < D; ++d) out[d] = 0.f;
        for (int i = 0; i < nq; ++i) {
            const float* E = cb + (size_t)i * K * D;
            const float* qin = (src0 && i == 0) ? x + (size_t)src0[n] * D : res;
            const float xn = sq_norm_4chains(qin, D);
            float best = -INFINITY;
            int bi = 0;
            for (int k = 0; k < K; ++k) {
                const float g = dot2_mfma_order(qin, E + (size_t)k * D, D);
                volatile float t1 = xn - g;
                volatile float t2 = t1 + enorm[(size_t)i * K + k];
                const float dist = -t2;
                if (dist > best) { best = dist; bi = k; }
            }
            codes[(size_t)i * N + n] = bi;
            const float* qv = E + (size_t)bi * D;
            for (int d = 0; d < D; ++d) {
                volatile float r = res[d] - qv[d];
                volatile float o = out[d] + qv[d];
                res[d] = r;
                out[d] = o;
            }
        }
        if (quant) memcpy(quant + (size_t)n * D, out, sizeof(float) * D);
    }
    free(enorm); free(res); free(out);
    return 0;
}

/* codes [N][nq] (the reference's token layout flattened over B*Tf) -> emb [N][D] */
int rvq_oracle_decode(const int64_t* codes, int N, int nq, int D, int K, const float* cb, float* emb) {
    for (int n = 0; n < N; ++n)
        for (int d = 0; d < D; ++d) {
            volatile float s = 0.f;
            for (int i = 0; i < nq; ++i) {
                int64_t idx = codes[(size_t)n * nq + i];
                if (idx < 0 || idx >= K) return -1;
                s = s + cb[((size_t)i * K + idx) * D + d];
            }
            emb[(size_t)n * D + d] = s;
        }
    return 0;
}
