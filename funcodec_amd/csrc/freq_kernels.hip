// gfx950 kernels that exist only for the STFT-domain codec (FreqCodec, funcodec/models/codec_freq.py:330-448).
//
// 2-D activations live in FREQUENCY-MAJOR layout [B][Fp][C][T] (Fp = F + 2*halo rows): one frequency row of one utterance is a
// contiguous [C][T] block, i.e. exactly what the 1-D implicit-GEMM conv kernel calls an utterance.  A Conv2d with kernel
// (kf, kt) and stride (sf, st) is then the 1-D conv (kt, st) over kf*C channels of the "virtual utterance" that starts at row
// fo*sf (conv_kernel.h, two-level batch addressing); the frequency padding (pad2d reflect, conv.py:100-119) is materialised as
// halo rows by halo_rows_kernel.  The STFT and its inverse are GEMMs on the same conv kernel (DFT matrix as conv weights over the
// hop-phase ("polyphase") view of the signal); what is left for this file is layout changes and pointwise math.
#include "kernels.h"

namespace fc {

static inline __host__ __device__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// xp[b][j][m] = xpad[m*hop + j], xpad = reflect-padded (n_fft/2 each side, torch.stft center=True) utterance, zero beyond it.
// (optionally / div[b]: the volume normalisation of _encode_frame, codec_freq.py:334-341)
__global__ __launch_bounds__(256) void polyphase_in_kernel(const float* __restrict__ wav, const float* __restrict__ div, int T, int hop,
                                                           int half, int Mp, float* __restrict__ xp) {
    const int b = blockIdx.z, j = blockIdx.y;
    const float d = div ? div[b] : 1.f;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < Mp; m += gridDim.x * 256) {
        const int n = m * hop + j - half;                 // index into the un-padded utterance
        float v = 0.f;
        if (n >= -half && n < T + half) {
            const int src = n < 0 ? -n : (n >= T ? 2 * (T - 1) - n : n);
            v = wav[(size_t)b * T + src];
            if (div) v = v / d;
        }
        xp[((size_t)b * hop + j) * Mp + m] = v;
    }
}

hipError_t launch_polyphase_in(const float* wav, const float* div, int B, int T, int hop, int n_fft, int Mp, float* xp, hipStream_t st) {
    hipLaunchKernelGGL(polyphase_in_kernel, dim3(cdiv(Mp, 256), hop, B), dim3(256), 0, st, wav, div, T, hop, n_fft / 2, Mp, xp);
    return hipGetLastError();
}

// STFT rows (re: 0..F-1, im: F..2F-1) [B][2F][Tp] -> features [B][halo + f][3][Tp] = (log(max(|X|, 1e-6)), X / max(|X|, 1e-6))
// (codec_freq.py:371-379)
__global__ __launch_bounds__(256) void stft_feats_kernel(const float* __restrict__ spec, int F, int Tp, long long spec_sB, int halo,
                                                         float* __restrict__ feats) {
    const int b = blockIdx.z, f = blockIdx.y;
    const float* re = spec + (size_t)b * spec_sB + (size_t)f * Tp;
    const float* im = re + (size_t)F * Tp;
    float* o = feats + (((size_t)b * (F + 2 * halo) + halo + f) * 3) * Tp;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < Tp; t += gridDim.x * 256) {
        const float a = re[t], c = im[t];
        const float mag = sqrtf(a * a + c * c);
        const float cl = fmaxf(mag, 1e-6f);
        o[t] = logf(cl);
        o[Tp + t] = a / cl;
        o[2 * (size_t)Tp + t] = c / cl;
    }
}

hipError_t launch_stft_feats(const float* spec, int B, int F, int Tp, long long spec_sB, int halo, float* feats, hipStream_t st) {
    hipLaunchKernelGGL(stft_feats_kernel, dim3(cdiv(Tp, 256), F, B), dim3(256), 0, st, spec, F, Tp, spec_sB, halo, feats);
    return hipGetLastError();
}

// halo rows of a frequency-major buffer [B][F + 2*halo][C][T]: reflect (row -i = row i, row F-1+i = row F-1-i; raw values, the
// consumer applies the per-channel affine to halo rows like to any other row) or zeros
__global__ __launch_bounds__(256) void halo_rows_kernel(float* __restrict__ buf, int F, int halo, int C, int T, int zero) {
    const int b = blockIdx.z, h = blockIdx.y;             // h in [0, 2*halo): first the rows above, then the rows below
    const int Fp = F + 2 * halo;
    const int dst = h < halo ? h : F + h;                 // padded row index
    const int i = h < halo ? halo - h : h - halo + 1;     // distance from the edge row
    const int srcf = h < halo ? i : F - 1 - i;            // un-padded source row
    float* d = buf + ((size_t)b * Fp + dst) * C * T;
    const float* s = buf + ((size_t)b * Fp + halo + (srcf < 0 ? 0 : (srcf >= F ? F - 1 : srcf))) * C * T;
    const bool z = zero || srcf < 0 || srcf >= F;
    const size_t n = (size_t)C * T;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) d[e] = z ? 0.f : s[e];
}

hipError_t launch_halo_rows(float* buf, int B, int F, int halo, int C, int T, int zero, hipStream_t st) {
    if (halo <= 0) return hipSuccess;
    int gx = cdiv(C * T, 256 * 8);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(halo_rows_kernel, dim3(gx, 2 * halo, B), dim3(256), 0, st, buf, F, halo, C, T, zero);
    return hipGetLastError();
}

// dst[b][hd + f][c][t] = [elu]( a0 * s0 + (s1 ? a1 * s1 : 0) ), sources / destination frequency-major with their own halos
__global__ __launch_bounds__(256) void combine2d_kernel(const float* __restrict__ s0, const float* __restrict__ aff0, int h0,
                                                        const float* __restrict__ s1, const float* __restrict__ aff1, int h1,
                                                        int elu, float alpha, int F, int C, int T, float* __restrict__ dst, int hd) {
    const int b = blockIdx.z, fc = blockIdx.y, f = fc / C, c = fc - f * C;
    float2 A0 = make_float2(1.f, 0.f), A1 = make_float2(1.f, 0.f);
    if (aff0) A0 = ((const float2*)aff0)[(size_t)b * C + c];
    if (s1 && aff1) A1 = ((const float2*)aff1)[(size_t)b * C + c];
    const float* r0 = s0 + (((size_t)b * (F + 2 * h0) + h0 + f) * C + c) * T;
    const float* r1 = s1 ? s1 + (((size_t)b * (F + 2 * h1) + h1 + f) * C + c) * T : r0;
    float* o = dst + (((size_t)b * (F + 2 * hd) + hd + f) * C + c) * T;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) {
        float v = fmaf(r0[t], A0.x, A0.y);
        if (s1) v = v + fmaf(r1[t], A1.x, A1.y);
        if (elu) { const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f); v = v > 0.f ? v : fmaf(e, alpha, -alpha); }
        o[t] = v;
    }
}

hipError_t launch_combine2d(const float* s0, const float* aff0, int h0, const float* s1, const float* aff1, int h1, int elu, float alpha,
                            int B, int F, int C, int T, float* dst, int hd, hipStream_t st) {
    int gx = cdiv(T, 256 * 4);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(combine2d_kernel, dim3(gx, F * C, B), dim3(256), 0, st, s0, aff0, h0, s1, aff1, h1, elu, alpha, F, C, T, dst, hd);
    return hipGetLastError();
}

// decoder output (raw, frequency-major [B][F + 2*halo][3][Tp], GroupNorm(1, 3) pending as aff[b][3]) -> spectrum rows for the
// inverse-STFT GEMM [B][2F][Tp]: softplus(mag) * (re, im)  (codec_freq.py:419-428; F.softplus: log1p(exp(x)), x for x > 20)
__global__ __launch_bounds__(256) void spec_from_dec_kernel(const float* __restrict__ dec, const float* __restrict__ aff, int F, int Tp,
                                                            int halo, float* __restrict__ spec) {
    const int b = blockIdx.z, f = blockIdx.y;
    const float2* A = (const float2*)aff + (size_t)b * 3;
    const float2 Am = aff ? A[0] : make_float2(1.f, 0.f), Ar = aff ? A[1] : make_float2(1.f, 0.f), Ai = aff ? A[2] : make_float2(1.f, 0.f);
    const float* r = dec + (((size_t)b * (F + 2 * halo) + halo + f) * 3) * Tp;
    float* ore = spec + ((size_t)b * 2 * F + f) * Tp;
    float* oim = ore + (size_t)F * Tp;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < Tp; t += gridDim.x * 256) {
        const float m = fmaf(r[t], Am.x, Am.y);
        const float sp = m > 20.f ? m : log1pf(expf(m));
        ore[t] = sp * fmaf(r[Tp + t], Ar.x, Ar.y);
        oim[t] = sp * fmaf(r[2 * (size_t)Tp + t], Ai.x, Ai.y);
    }
}

hipError_t launch_spec_from_dec(const float* dec, const float* aff, int B, int F, int Tp, int halo, float* spec, hipStream_t st) {
    hipLaunchKernelGGL(spec_from_dec_kernel, dim3(cdiv(Tp, 256), F, B), dim3(256), 0, st, dec, aff, F, Tp, halo, spec);
    return hipGetLastError();
}

// overlap-added windowed frames in hop-phase layout ypoly[b][j][m] (sample m*hop + j of the padded signal) -> waveform: divide by
// the window envelope sum_t w^2[n - t*hop] (torch.istft), drop the n_fft/2 centre padding, x scale[b], first out_len samples
__global__ __launch_bounds__(256) void istft_finish_kernel(const float* __restrict__ ypoly, const float* __restrict__ win2, int hop, int n_fft,
                                                           int Mp, int Tp, const float* __restrict__ mul, int out_len, float* __restrict__ wav) {
    const int b = blockIdx.y;
    const float s = mul ? mul[b] : 1.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < out_len; i += gridDim.x * 256) {
        const int n = i + n_fft / 2;                      // index in the padded signal
        const int m = n / hop, j = n - m * hop;
        float env = 0.f;                                  // frames t with 0 <= n - t*hop < n_fft, ascending t like torch's fold
        int t_lo = (n - n_fft + hop) / hop;
        if (t_lo < 0) t_lo = 0;
        int t_hi = n / hop;
        if (t_hi > Tp - 1) t_hi = Tp - 1;
        for (int t = t_lo; t <= t_hi; ++t) env += win2[n - t * hop];
        const float v = ypoly[((size_t)b * hop + j) * Mp + m];
        wav[(size_t)b * out_len + i] = (env > 1e-11f ? v / env : v) * s;
    }
}

hipError_t launch_istft_finish(const float* ypoly, const float* win2, int B, int hop, int n_fft, int Mp, int Tp, const float* mul, int out_len,
                               float* wav, hipStream_t st) {
    hipLaunchKernelGGL(istft_finish_kernel, dim3(cdiv(out_len, 256 * 4), B), dim3(256), 0, st, ypoly, win2, hop, n_fft, Mp, Tp, mul, out_len, wav);
    return hipGetLastError();
}

}  // namespace fc
