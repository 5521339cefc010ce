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
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// STFT rows (re: 0..F-1, im: F..2F-1) [B][2F][Tp] -> features, frequency-major [B][halo + f][C][Tp]:
//   C = 3 (codec_domain mag_phase, codec_freq.py:371-379): log(max(|X|, 1e-6)), X / max(|X|, 1e-6)
//   C = 2 (codec_domain mag_angle, codec_freq.py:356-364): log(max(|X|, 1e-6)), torch.angle(X) = atan2(im, re)
template <int C>
__global__ __launch_bounds__(256) void stft_feats_kernel(const float* __restrict__ spec, int F, int Tp, long long spec_sB, int halo,
                                                         float* __restrict__ feats) {
    const int b = blockIdx.z, f = blockIdx.y;
    const float* re = spec + (size_t)b * spec_sB + (size_t)f * Tp;
    const float* im = re + (size_t)F * Tp;
    float* o = feats + (((size_t)b * (F + 2 * halo) + halo + f) * C) * Tp;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < Tp; t += gridDim.x * 256) {
        const float a = re[t], c = im[t];
        const float mag = sqrtf(a * a + c * c);
        const float cl = fmaxf(mag, 1e-6f);
        o[t] = logf(cl);
        if (C == 3) {
            o[Tp + t] = a / cl;
            o[2 * (size_t)Tp + t] = c / cl;
        } else {
            o[Tp + t] = atan2f(c, a);
        }
    }
}

hipError_t launch_stft_feats(const float* spec, int B, int F, int Tp, long long spec_sB, int halo, int C, float* feats, hipStream_t st) {
    if (C == 3) hipLaunchKernelGGL(stft_feats_kernel<3>, dim3(cdiv(Tp, 256), F, B), dim3(256), 0, st, spec, F, Tp, spec_sB, halo, feats);
    else if (C == 2) hipLaunchKernelGGL(stft_feats_kernel<2>, dim3(cdiv(Tp, 256), F, B), dim3(256), 0, st, spec, F, Tp, spec_sB, halo, feats);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// test hooks (fc_debug_freq_features): the feature tensor in the reference's layout [B][C][F][Tp] <-> the engine's [B][halo + f][C][Tp]
__global__ __launch_bounds__(256) void feats_relayout_kernel(float* __restrict__ eng, float* __restrict__ ref, int C, int F, int Tp, int halo, int to_ref) {
    const int b = blockIdx.z, fc = blockIdx.y, f = fc / C, c = fc - f * C;
    float* e = eng + (((size_t)b * (F + 2 * halo) + halo + f) * C + c) * Tp;
    float* r = ref + (((size_t)b * C + c) * F + f) * Tp;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < Tp; t += gridDim.x * 256) {
        if (to_ref) r[t] = e[t];
        else e[t] = r[t];
    }
}
hipError_t launch_feats_relayout(float* eng, float* ref, int B, int C, int F, int Tp, int halo, int to_ref, hipStream_t st) {
    hipLaunchKernelGGL(feats_relayout_kernel, dim3(cdiv(Tp, 256), F * C, B), dim3(256), 0, st, eng, ref, C, F, Tp, halo, to_ref);
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
    if ((n & 3) == 0) {                                   // rows of C * T floats start 16-byte aligned when C * T is a multiple of 4
        const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n / 4; e += (size_t)gridDim.x * 256)
            ((f32x4*)d)[e] = z ? zz : ((const f32x4*)s)[e];
        return;
    }
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) d[e] = z ? 0.f : s[e];
}

hipError_t launch_halo_rows(float* buf, int B, int F, int halo, int C, int T, int zero, hipStream_t st) {
    if (halo <= 0) return hipSuccess;
    int gx = cdiv(C * T, 256 * 8);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(halo_rows_kernel, dim3(gx, 2 * halo, B), dim3(256), 0, st, buf, F, halo, C, T, zero);
    return hipGetLastError();
}

// dst[b][hd + f][c][t] = [elu]( a0 * s0 + (s1 ? a1 * s1 : 0) ), sources / destination frequency-major with their own halos.
// Round 4: 16-byte accesses (a lane owns 4 consecutive samples of a row; rows of T = 1001 floats start at every alignment, hence the
// 4-byte-aligned vector type).  These materialisation passes are 13 % of a FreqCodec gr1 call and were streaming dword by dword.
__global__ __launch_bounds__(256) void combine2d_kernel(const float* __restrict__ s0, const float* __restrict__ aff0, int h0,
                                                        const float* __restrict__ s1, const float* __restrict__ aff1, int h1,
                                                        int elu, float alpha, int F, int C, int T, float* __restrict__ dst, int hd, int halo_mode) {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const int b = blockIdx.z, fc = blockIdx.y, f = fc / C, c = fc - f * C;
    // halo rows of dst written here instead of by a halo_rows launch (round 4): mode 1 = reflect (row -i = row i, row F-1+i = row F-1-i; the
    // row's own workgroup stores it a second time), mode 2 = zeros (the edge rows' workgroups clear them).  Offsets in rows relative to row f.
    int m0 = 0, m1 = 0;
    if (halo_mode == 1) {
        if (f >= 1 && f <= hd) m0 = -2 * f;
        if (f <= F - 2 && f >= F - 1 - hd) m1 = 2 * (F - 1 - f);
    }
    float2 A0 = make_float2(1.f, 0.f), A1 = make_float2(1.f, 0.f);
    if (aff0) A0 = ((const float2*)aff0)[(size_t)b * C + c];
    if (s1 && aff1) A1 = ((const float2*)aff1)[(size_t)b * C + c];
    const float* r0 = s0 + (((size_t)b * (F + 2 * h0) + h0 + f) * C + c) * T;
    const float* r1 = s1 ? s1 + (((size_t)b * (F + 2 * h1) + h1 + f) * C + c) * T : r0;
    float* o = dst + (((size_t)b * (F + 2 * hd) + hd + f) * C + c) * T;
    auto act = [&](float x0, float x1) __attribute__((always_inline)) {
        float v = fmaf(x0, A0.x, A0.y);
        if (s1) v = v + fmaf(x1, A1.x, A1.y);
        if (elu) { const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f); v = v > 0.f ? v : fmaf(e, alpha, -alpha); }
        return v;
    };
    for (int t = 4 * (blockIdx.x * 256 + threadIdx.x); t < T; t += 4 * gridDim.x * 256) {
        if (t + 3 < T) {
            const f32x4 x0 = *(const f32x4u*)(r0 + t);
            const f32x4 x1 = s1 ? (f32x4)(*(const f32x4u*)(r1 + t)) : x0;
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act(x0[j], x1[j]);
            *(f32x4u*)(o + t) = v;
            if (m0) *(f32x4u*)(o + (long long)m0 * C * T + t) = v;
            if (m1) *(f32x4u*)(o + (long long)m1 * C * T + t) = v;
        } else {
            for (int j = 0; t + j < T; ++j) {
                const float v = act(r0[t + j], r1[t + j]);
                o[t + j] = v;
                if (m0) o[(long long)m0 * C * T + t + j] = v;
                if (m1) o[(long long)m1 * C * T + t + j] = v;
            }
        }
        if (halo_mode == 2) {
            for (int i = 1; i <= hd; ++i) {
                if (f == 0) for (int j = 0; j < 4 && t + j < T; ++j) o[-(long long)i * C * T + t + j] = 0.f;
                if (f == F - 1) for (int j = 0; j < 4 && t + j < T; ++j) o[(long long)i * C * T + t + j] = 0.f;
            }
        }
    }
}

hipError_t launch_combine2d(const float* s0, const float* aff0, int h0, const float* s1, const float* aff1, int h1, int elu, float alpha,
                            int B, int F, int C, int T, float* dst, int hd, hipStream_t st, int halo_mode) {
    int gx = cdiv(T, 256 * 4);
    if (gx < 1) gx = 1;
    if (halo_mode == 1 && F <= hd) return hipErrorInvalidValue;       // a reflected row would have no source: the caller keeps halo_rows
    hipLaunchKernelGGL(combine2d_kernel, dim3(gx, F * C, B), dim3(256), 0, st, s0, aff0, h0, s1, aff1, h1, elu, alpha, F, C, T, dst, hd, hd > 0 ? halo_mode : 0);
    return hipGetLastError();
}

// decoder output (raw, frequency-major [B][F + 2*halo][C][Tp], GroupNorm(1, C) pending as aff[b][C]) -> spectrum rows for the
// inverse-STFT GEMM [B][2F][Tp]  (F.softplus: log1p(exp(x)), x for x > 20):
//   C = 3 (mag_phase, codec_freq.py:419-428): softplus(mag) * (re, im)
//   C = 2 (mag_angle, codec_freq.py:426-434): mag = softplus(o0), angle = sin(o1) * pi, spectrum = (cos(angle) mag, sin(angle) mag)
template <int C>
__global__ __launch_bounds__(256) void spec_from_dec_kernel(const float* __restrict__ dec, const float* __restrict__ aff, int F, int Tp,
                                                            int halo, float* __restrict__ spec) {
    const int b = blockIdx.z, f = blockIdx.y;
    const float2* A = (const float2*)aff + (size_t)b * C;
    const float2 one = make_float2(1.f, 0.f);
    const float2 Am = aff ? A[0] : one, Ar = aff ? A[1] : one, Ai = (aff && C == 3) ? A[C - 1] : one;
    const float* r = dec + (((size_t)b * (F + 2 * halo) + halo + f) * C) * Tp;
    float* ore = spec + ((size_t)b * 2 * F + f) * Tp;
    float* oim = ore + (size_t)F * Tp;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < Tp; t += gridDim.x * 256) {
        const float m = fmaf(r[t], Am.x, Am.y);
        const float sp = m > 20.f ? m : log1pf(expf(m));
        const float o1 = fmaf(r[Tp + t], Ar.x, Ar.y);
        if (C == 3) {
            ore[t] = sp * o1;
            oim[t] = sp * fmaf(r[2 * (size_t)Tp + t], Ai.x, Ai.y);
        } else {
            const float ang = sinf(o1) * 3.14159265358979323846f;      // torch.pi rounded to fp32 by the multiplication's operand
            ore[t] = cosf(ang) * sp;
            oim[t] = sinf(ang) * sp;
        }
    }
}

hipError_t launch_spec_from_dec(const float* dec, const float* aff, int B, int F, int Tp, int halo, int C, float* spec, hipStream_t st) {
    if (C == 3) hipLaunchKernelGGL(spec_from_dec_kernel<3>, dim3(cdiv(Tp, 256), F, B), dim3(256), 0, st, dec, aff, F, Tp, halo, spec);
    else if (C == 2) hipLaunchKernelGGL(spec_from_dec_kernel<2>, dim3(cdiv(Tp, 256), F, B), dim3(256), 0, st, dec, aff, F, Tp, halo, spec);
    else return hipErrorInvalidValue;
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

// =================================================================================================
// Grouped Conv2d with a few channels per group (conv_group_ratio = 1: 2 or 4 in, 2 or 4 out; seanet_encoder.py:224,234,321):
// 18 .. 64 multiply-adds per output -- nothing for the matrix cores, the layer is bound by HBM and the prologue's VALU work.
// Direct form, no LDS slab, no barrier: one workgroup = (utterance, output frequency row, group, 1024 output columns); a lane owns 4
// consecutive output columns of all OPG output channels of the group, loads the 3*ST + KT input columns it needs per (frequency tap,
// input channel) with 16-byte pieces (straight-line, clamped; lanes at a padded edge gather by index), applies the prologue once per
// loaded sample and accumulates OPG x 4 outputs.  The group's weights (<= 128 floats) sit in LDS and are read as broadcasts.
// Same contract as the implicit-GEMM kernel: raw output + bias, fp64 (sum, sum of squares) partial per workgroup.
// =================================================================================================
struct GConvArgs {
    const float *src0, *aff0, *src1, *aff1;      // [B][Fin + 2h][C][Tin] (pointing at the first row the fo = 0 window reads), affines [B][C][2]
    const float *w, *bias;                       // torch layout [M][CPG][KF][KT], [M]
    float* out;                                  // [B][Fo + 2ho][M][Tout], pointing at row ho
    double* partials;                            // [B][Fo][G][ttiles][2]
    int C, M, Tin, Tout, Fo, G, sf;
    int padL, Leff, elu;
    float alpha;
    long long in_sB, in_sF, out_sB, out_sF;      // floats between utterances / frequency rows
    int out_halo;                                // > 0: reflected halo rows of the output written by the rows' own workgroups (Fo > out_halo)
};

// Where the group's weights come from: (KF >= 4) the strided 8-row layers read them as uniform SCALAR loads -- staged in LDS the compiler
// preloaded all 128 .. 256 of them into vector registers (250 - 256 registers, 1 - 2 waves per SIMD for a latency-bound streaming kernel);
// from SGPRs the kernels take 128 / 208 registers: 0.99 -> 0.64 and 0.77 -> 0.53 ms per 32-utterance call.  The 3 x 3 and 1 x 1 layers are
// VALU-bound and measured 8 % / 4 % SLOWER that way (s_load waits in the arithmetic), they keep the LDS broadcasts.  -1 = by shape.
// (All 72 weights of the 3 x 3 kernel pinned in SGPRs: 65 SGPR spills; volatile LDS reads at every use: 256 registers.  Neither was run.)
#ifndef FC_GCONV_WSCALAR
#define FC_GCONV_WSCALAR -1
#endif
#ifndef FC_GCONV_WAVES
#define FC_GCONV_WAVES 0
#endif
#ifndef FC_GCONV_FASTEDGE
#define FC_GCONV_FASTEDGE 1
#endif
#ifndef FC_GCONV_ABL
#define FC_GCONV_ABL 0        // profiling builds (results are garbage): 1 no ELU arithmetic, 2 a quarter of the FMAs
#endif
#if FC_GCONV_WAVES > 0
#define FC_GCONV_ATTR __attribute__((amdgpu_waves_per_eu(FC_GCONV_WAVES, FC_GCONV_WAVES)))
#else
#define FC_GCONV_ATTR
#endif
template <int CPG, int OPG, int KF, int KT, int ST, bool DUAL, int FO, bool NEEDMASK>
__global__ __launch_bounds__(256) FC_GCONV_ATTR void gconv2d_kernel(const GConvArgs p) {
    // FO = output frequency rows per lane (rows fo0, fo0 + 1): the KF + (FO - 1) SF input rows they read are loaded and activated once
    // instead of FO x KF times (3x3: 4 rows instead of 6; 8-row stride-4 layers: 12 instead of 16)
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    constexpr int NE = 3 * ST + KT;              // input columns behind 4 outputs
    constexpr int NV = (NE + 3) / 4;             // 16-byte pieces
    constexpr int NW = OPG * CPG * KF * KT;
    __shared__ float wsh[NW];
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = blockIdx.x, g = blockIdx.y, z = blockIdx.z;
    const int FoP = (p.Fo + FO - 1) / FO;        // row groups per utterance
    const int b = z / FoP, fo0 = (z - b * FoP) * FO;
    const int sf = p.sf;
    constexpr bool WSCALAR = FC_GCONV_WSCALAR < 0 ? KF >= 4 : FC_GCONV_WSCALAR != 0;
    const float* wgp = p.w + (size_t)g * NW;      // uniform per workgroup: scalar loads
    if (!WSCALAR) {
        for (int i = tid; i < NW; i += 256) wsh[i] = wgp[i];
        __syncthreads();
        wgp = wsh;
    }
    const int n0 = tile * 1024 + 4 * tid;        // first output column of this lane
    const int q0 = n0 * ST - p.padL;             // source index (before reflection) of its first input column
    // FASTEDGE (3 x 3, stride 1, rows longer than the padding; round 4): NO clamped addresses and NO per-element gathers for the lanes at the
    // row ends -- those gathers (address registers, ~100 divergent regions) cost 60 of the kernel's 232 registers.  Every lane loads its
    // window where it lies: the floats before column 0 / behind column T - 1 are the neighbouring channel row's (the activation buffers are
    // contiguous; the workspace starts 256 bytes in and ends with 4 KiB of slack), i.e. finite garbage, and the only two garbage columns a VALID
    // output reads are replaced in registers by their reflections: column -1 by column 1 (the lane with q0 = -1: element 0 <- element 2) and
    // column T by column T - 2 (the lane that owns output T - 1: element eT = T - q0 in 2 .. 5 <- element eT - 2); causal nets pad two
    // columns on the left and none on the right (elements 0, 1 <- elements 4, 3).
    constexpr bool FASTEDGE = !NEEDMASK && ST == 1 && ((KF == 3 && KT == 3) || KT == 2) && FC_GCONV_FASTEDGE;
    const bool vec_ok = FASTEDGE || (q0 >= 0 && q0 + 4 * NV <= p.Tin);
    const int qsafe = FASTEDGE ? q0 : (q0 < 0 ? 0 : (q0 + 4 * NV <= p.Tin ? q0 : (p.Tin >= 4 * NV ? p.Tin - 4 * NV : 0)));
    const int eT = p.Tin - q0;                   // FASTEDGE: element index of column T in this lane's window
    int esrc[NE]; unsigned emask = 0;
    {
        const int refl = 2 * (p.Leff - 1);
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int q = q0 + j;
            int src = q < 0 ? -q : q;
            src = src >= p.Leff ? refl - src : src;
            const bool ok = q >= -p.padL && src >= 0 && src < p.Tin;
            esrc[j] = ok ? src : 0;
            emask |= (ok ? 1u : 0u) << j;
        }
    }
    const unsigned vmask = vec_ok ? (1u << NE) - 1u : emask;
    const int row_max = (p.Fo - 1) * sf + KF - 1;    // last input row any valid output row reads (a dangling second row is clamped onto it)
    const size_t in_b = (size_t)b * p.in_sB + (size_t)(g * CPG) * p.Tin;
    const float2* a0 = p.aff0 ? (const float2*)p.aff0 + (size_t)b * p.C + g * CPG : nullptr;
    const float2* a1 = (DUAL && p.aff1) ? (const float2*)p.aff1 + (size_t)b * p.C + g * CPG : nullptr;
    float acc[FO][OPG][4];
#pragma unroll
    for (int f = 0; f < FO; ++f)
#pragma unroll
        for (int o = 0; o < OPG; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[f][o][j] = 0.f;
    const bool live = n0 < p.Tout;
    // Round 4: the select that zeroes columns without a source is compiled in only for rows not longer than the padding (NEEDMASK, chosen by
    // the launcher: Leff != Tin, the reference zero-extends those before reflecting) -- for longer rows every column a VALID output reads has
    // a reflected source, and the columns without one feed discarded outputs only.  (Both variants behind a branch in one kernel, or the edge
    // lanes' gathers as one divergent region per row instead of one per element: 256 registers, one wave per SIMD -- not run.  A ROLLED form --
    // a real loop over row pairs with the frequency tap as a run-time index into a zero-padded LDS weight table, so that the 72 weights are
    // not held in vector registers: 196 registers without the edge gathers, 256 with them, 107 spills at a forced 168 -- not run either: the
    // per-element gather addresses, not the weights, are what fills the register file.)
    if (live) {
        // the stride between the FO rows' windows is a run-time value; the reference's 2-D nets use sf == KF / 2 for strided layers and
        // sf == 1 otherwise, which is what the (row, output) -> tap table below is unrolled for
        constexpr int NR = KF + (FO - 1) * (KF >= 4 ? KF / 2 : 1);
        // round 3: the NEXT input row's loads are requested before the current row is activated and multiplied (two register sets): as
        // one straight-line block per row the loads of row r + 1 were issued only after row r's ~150 VALU instructions had retired
        // FASTEDGE: the group's channels in passes of CP = 2 -- a REAL loop (straight-line over all 4 channels, without the gather regions
        // as scheduling fences, hipcc interleaves everything and takes 254 registers = one wave per SIMD): per pass 4 rows x 2 channels,
        // one row ahead, the pass's 36 weights; the accumulators carry over.
        constexpr int CP = (FASTEDGE && CPG == 4) ? 2 : CPG;
        f32x4 rbuf0[2][CP][NV], rbuf1[2][DUAL ? CP : 1][NV];
#pragma unroll CP == CPG ? 4 : 1
        for (int c0 = 0; c0 < CPG; c0 += CP) {
            auto load_row = [&](int r, int slot) __attribute__((always_inline)) {
                int rr = fo0 * sf + r;
                rr = rr > row_max ? row_max : rr;
#pragma unroll
                for (int ci = 0; ci < CP; ++ci) {   // straight-line loads of the whole input row
                    const size_t off = in_b + (size_t)rr * p.in_sF + (size_t)(c0 + ci) * p.Tin + qsafe;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        rbuf0[slot][ci][v] = *(const f32x4u*)(p.src0 + off + 4 * v);
                        if (DUAL) rbuf1[slot][ci][v] = *(const f32x4u*)(p.src1 + off + 4 * v);
                    }
                }
            };
            load_row(0, 0);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                int rr = fo0 * sf + r;
                rr = rr > row_max ? row_max : rr;
                if (r + 1 < NR) load_row(r + 1, (r + 1) & 1);
                f32x4 (&r0)[CP][NV] = rbuf0[r & 1];
                f32x4 (&r1)[DUAL ? CP : 1][NV] = rbuf1[r & 1];
#pragma unroll
                for (int ci = 0; ci < CP; ++ci) {
                    const int cc = c0 + ci;          // channel inside the group
                    const float2 A = a0 ? a0[cc] : make_float2(1.f, 0.f);
                    const float2 A1 = a1 ? a1[cc] : make_float2(1.f, 0.f);
                    float x[NE];
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        float v0 = r0[ci][j >> 2][j & 3], v1 = DUAL ? r1[ci][j >> 2][j & 3] : 0.f;
                        if (!vec_ok) {               // padded edge: gather by index
                            const size_t off = in_b + (size_t)rr * p.in_sF + (size_t)cc * p.Tin + esrc[j];
                            v0 = p.src0[off];
                            if (DUAL) v1 = p.src1[off];
                        }
                        float v = fmaf(v0, A.x, A.y);
                        if (DUAL) v = v + fmaf(v1, A1.x, A1.y);
                        if (p.elu && !(FC_GCONV_ABL & 1)) { const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f); v = v > 0.f ? v : fmaf(e, p.alpha, -p.alpha); }
                        x[j] = (!NEEDMASK || ((vmask >> j) & 1u)) ? v : 0.f;
                    }
                    if (FASTEDGE) {                  // reflections of the columns outside the row that valid outputs read
                        // left: column -k <- column k, i.e. element padL - k <- element padL + k (the lane with q0 = -padL; padL = 1, or 2 = causal)
                        if (q0 < 0) {
                            x[0] = p.padL == 1 ? x[2] : x[4 < NE ? 4 : 0];
                            if (KT == 3 && p.padL == 2) x[1] = x[3];
                        }
                        // right (3 taps, non-causal only: one padded column): column T <- column T - 2 in the lane that owns output T - 1;
                        // 2 taps (the strided 8-row layers' time axis): one padded column on the left, none on the right
                        if (KT == 3 && p.padL == 1) {
#pragma unroll
                            for (int e = 2; e < NE; ++e)
                                if (eT == e) x[e] = x[e - 2];
                        }
                    }
#pragma unroll
                    for (int f = 0; f < FO; ++f) {
                        constexpr int SFC = KF >= 4 ? KF / 2 : 1;          // compile-time row stride between the FO windows (== p.sf, checked by the launcher)
                        const int a = r - f * SFC;                         // frequency tap of input row r for output row f
                        if (a < 0 || a >= KF) continue;
#pragma unroll
                        for (int o = 0; o < OPG; ++o)
#pragma unroll
                            for (int kk = 0; kk < KT; ++kk) {
                                const float wv = wgp[((o * CPG + cc) * KF + a) * KT + kk];
                                if (FC_GCONV_ABL & 2) { acc[f][o][0] += wv * x[kk]; continue; }       // profiling build: a quarter of the FMAs
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[f][o][j] = fmaf(wv, x[j * ST + kk], acc[f][o][j]);
                            }
                    }
                }
            }
        }
    }
    float s1v = 0.f, s2v = 0.f;
#pragma unroll
    for (int f = 0; f < FO; ++f) {
        const int fo = fo0 + f;
        if (fo >= p.Fo) continue;                // dangling second row of an odd row count
#pragma unroll
        for (int o = 0; o < OPG; ++o) {
            const int m = g * OPG + o;
            const float bm = p.bias[m];
            float ov[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ov[j] = acc[f][o][j] + bm;
                if (live && n0 + j < p.Tout) { s1v += ov[j]; s2v = fmaf(ov[j], ov[j], s2v); }
            }
            if (!live) continue;
            float* orow = p.out + (size_t)b * p.out_sB + (size_t)fo * p.out_sF + (size_t)m * p.Tout + n0;
            // reflected halo rows of the output (row -i = row i, row Fo-1+i = row Fo-1-i): this row's second / third store
            const int h = p.out_halo;
            const long long mir0 = (h && fo >= 1 && fo <= h) ? -2ll * fo * p.out_sF : 0;
            const long long mir1 = (h && fo <= p.Fo - 2 && fo >= p.Fo - 1 - h) ? 2ll * (p.Fo - 1 - fo) * p.out_sF : 0;
            if (n0 + 3 < p.Tout) {
                const f32x4 o4 = {ov[0], ov[1], ov[2], ov[3]};
                *(f32x4u*)orow = o4;
                if (mir0) *(f32x4u*)(orow + mir0) = o4;
                if (mir1) *(f32x4u*)(orow + mir1) = o4;
            } else
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n0 + j < p.Tout) {
                        orow[j] = ov[j];
                        if (mir0) orow[mir0 + j] = ov[j];
                        if (mir1) orow[mir1 + j] = ov[j];
                    }
        }
    }
    if (p.partials) {
        double d1 = (double)s1v, d2 = (double)s2v;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            d1 += __shfl_xor(d1, off, 64);
            d2 += __shfl_xor(d2, off, 64);
        }
        if (lane == 0) { red[0][wid] = d1; red[1][wid] = d2; }
        __syncthreads();
        if (tid == 0) {
            const size_t slot = ((((size_t)b * FoP + (fo0 / FO)) * p.G + g) * gridDim.x + tile) * 2;
            p.partials[slot] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
            p.partials[slot + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        }
    }
}


#ifdef FC_AB_KNOBS      // measured slower than the direct form (see gconv2d_lds3): compiled only into A / B builds
// -------------------------------------------------------------------------------------------------
// Round 4: the 3 x 3 layers (stride 1) through LDS.  The direct form above activates every loaded input sample once per lane that reads
// it: 4-column lanes overlap by 2 columns (1.5x) and 2-row lanes by 2 rows (2x), 3 activations per sample, and the layer is bound by
// v_exp_f32 + the affine / select work around it, not by HBM.  Here one workgroup owns an 8-row x 256-column output
// tile of one (utterance, group): the (8 + 2) x (256 + 2) x CPG input patch is loaded once (all 16-byte pieces of a thread in flight
// before the first use), affine'd / summed / ELU'd ONCE per sample (1.26 activations per sample with the halo) into LDS, and every lane
// computes its 2 rows x 4 columns x OPG outputs from LDS reads (one 16-byte + one 8-byte read per (row, channel)); the group's weights are
// uniform scalar loads.  Same contract: raw output + bias, one fp64 (sum, sum of squares) partial per workgroup.
// -------------------------------------------------------------------------------------------------
constexpr int G3_RF = 8, G3_TN = 256, G3_PW = 264;       // output rows / columns per workgroup, LDS row stride (258 used, 16-byte multiple)

template <int CPG, int OPG, bool DUAL>
__global__ __launch_bounds__(256) void gconv2d_3x3_lds_kernel(const GConvArgs p) {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int PR = G3_RF + 2;                        // patch rows
    constexpr int NPC = (G3_TN + 2 + 3) / 4;             // 16-byte pieces per patch row (65: columns 0 .. 259)
    constexpr int NP = NPC * PR * CPG;                   // pieces of the patch
    constexpr int NIT = (NP + 255) / 256;
    constexpr int NW = OPG * CPG * 9;
    extern __shared__ __attribute__((aligned(16))) float patch[];       // [CPG][PR][G3_PW]
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = blockIdx.x, g = blockIdx.y, z = blockIdx.z;
    const int FoT = (p.Fo + G3_RF - 1) / G3_RF;
    const int b = z / FoT, fo0 = (z - b * FoT) * G3_RF;
    const int n0 = tile * G3_TN;                         // first output column of the tile; patch column c is source column n0 - padL + c
    const int row_max = p.Fo + 1;                        // last input row a valid output row reads (dangling rows are clamped onto it)
    const size_t in_b = (size_t)b * p.in_sB + (size_t)(g * CPG) * p.Tin;
    const float2* a0 = p.aff0 ? (const float2*)p.aff0 + (size_t)b * p.C + g * CPG : nullptr;
    const float2* a1 = (DUAL && p.aff1) ? (const float2*)p.aff1 + (size_t)b * p.C + g * CPG : nullptr;
    const int refl = 2 * (p.Leff - 1);
    // ---- stage: every piece of this thread is requested before the first one is used
    f32x4 v0[NIT], v1[DUAL ? NIT : 1];
    int meta[NIT];                                       // (ci, rr, i) of the piece, -1 = none
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 256 + tid;
        const int ic = idx < NP ? idx : NP - 1;
        const int i = ic % NPC, rem = ic / NPC, rr = rem % PR, ci = rem / PR;
        meta[it] = idx < NP ? ic : -1;
        int rm = fo0 + rr;
        rm = rm > row_max ? row_max : rm;
        const size_t off = in_b + (size_t)rm * p.in_sF + (size_t)ci * p.Tin;
        const int q0 = n0 - p.padL + 4 * i;
        if (q0 >= 0 && q0 + 3 < p.Tin) {                 // interior piece: one 16-byte load per source
            v0[it] = *(const f32x4u*)(p.src0 + off + q0);
            if (DUAL) v1[it] = *(const f32x4u*)(p.src1 + off + q0);
        } else {                                         // padded edge: reflected gather, zeros outside the padded signal
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = q0 + j;
                int src = q < 0 ? -q : q;
                src = src >= p.Leff ? refl - src : src;
                const bool ok = q >= -p.padL && src >= 0 && src < p.Tin;
                v0[it][j] = ok ? p.src0[off + src] : 0.f;
                if (DUAL) v1[it][j] = ok ? p.src1[off + src] : 0.f;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (meta[it] < 0) continue;
        const int ic = meta[it];
        const int i = ic % NPC, rem = ic / NPC, rr = rem % PR, ci = rem / PR;
        const float2 A = a0 ? a0[ci] : make_float2(1.f, 0.f);
        const float2 A1 = a1 ? a1[ci] : make_float2(1.f, 0.f);
        const int q0 = n0 - p.padL + 4 * i;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + j;
            int src = q < 0 ? -q : q;
            src = src >= p.Leff ? refl - src : src;
            const bool ok = q >= -p.padL && src >= 0 && src < p.Tin;
            float v = fmaf(v0[it][j], A.x, A.y);
            if (DUAL) v = v + fmaf(v1[it][j], A1.x, A1.y);
            if (p.elu) { const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f); v = v > 0.f ? v : fmaf(e, p.alpha, -p.alpha); }
            o[j] = ok ? v : 0.f;
        }
        *(f32x4*)(patch + ((size_t)ci * PR + rr) * G3_PW + 4 * i) = o;
    }
    __syncthreads();
    // ---- compute: wave w -> output rows 2w, 2w + 1 of the tile; lane -> 4 columns
    const float* wg = p.w + (size_t)g * NW;              // uniform: scalar loads
    float acc[2][OPG][4];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int o = 0; o < OPG; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[f][o][j] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CPG; ++ci) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {                    // input rows 2w + a of the patch
            const float* pr = patch + ((size_t)ci * PR + 2 * wid + a) * G3_PW + 4 * lane;
            const f32x4 xa = *(const f32x4*)pr;
            const f32x2 xb = *(const f32x2*)(pr + 4);
            const float x[6] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1]};
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int tap = a - f;                   // frequency tap of this input row for output row 2w + f
                if (tap < 0 || tap > 2) continue;
#pragma unroll
                for (int o = 0; o < OPG; ++o)
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        const float wv = wg[((o * CPG + ci) * 3 + tap) * 3 + kk];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[f][o][j] = fmaf(wv, x[j + kk], acc[f][o][j]);
                    }
            }
        }
    }
    // ---- epilogue
    const int nl = n0 + 4 * lane;
    const bool live = nl < p.Tout;
    float s1v = 0.f, s2v = 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int fo = fo0 + 2 * wid + f;
        if (fo >= p.Fo) continue;
#pragma unroll
        for (int o = 0; o < OPG; ++o) {
            const int m = g * OPG + o;
            const float bm = p.bias[m];
            float ov[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ov[j] = acc[f][o][j] + bm;
                if (live && nl + j < p.Tout) { s1v += ov[j]; s2v = fmaf(ov[j], ov[j], s2v); }
            }
            if (!live) continue;
            float* orow = p.out + (size_t)b * p.out_sB + (size_t)fo * p.out_sF + (size_t)m * p.Tout + nl;
            if (nl + 3 < p.Tout) *(f32x4u*)orow = (f32x4){ov[0], ov[1], ov[2], ov[3]};
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) if (nl + j < p.Tout) orow[j] = ov[j];
        }
    }
    if (p.partials) {
        double d1 = (double)s1v, d2 = (double)s2v;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            d1 += __shfl_xor(d1, off, 64);
            d2 += __shfl_xor(d2, off, 64);
        }
        if (lane == 0) { red[0][wid] = d1; red[1][wid] = d2; }
        __syncthreads();
        if (tid == 0) {
            const size_t slot = ((((size_t)b * FoT + (fo0 / G3_RF)) * p.G + g) * gridDim.x + tile) * 2;
            p.partials[slot] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
            p.partials[slot + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        }
    }
}

#endif

static bool gconv2d_lds3(int kf, int kt, int st) {
    // OFF by default.  Measured on MI355X (freqmpgr1, 64 x 10 s, both forms in one call, all 42 FreqCodec tests green with either): the 8
    // launches of the 3 x 3 class take 3.04 ms through LDS against 2.58 ms direct.  The activations did drop from 3 to 1.26 per sample, but
    // the layer is not VALU-bound enough for that to decide: a workgroup here loads, activates, barriers and then computes -- phases that
    // only other workgroups can overlap, and 163 registers + 42 KB of LDS leave 3 of them per CU -- while the direct form has no barrier
    // and every wave streams on its own.  What would beat it is a persistent tile loop with a double-buffered patch; FC_GCONV_LDS3=1
    // selects this kernel for A / B runs.
#ifdef FC_AB_KNOBS
    static const int env = ab_knob("FC_GCONV_LDS3", 0);
    return env && kf == 3 && kt == 3 && st == 1;
#else
    (void)kf; (void)kt; (void)st;
    return false;
#endif
}

bool gconv2d_ok(int cpg, int opg, int kf, int kt, int st) {
    const bool shape = (kf == 1 && kt == 1 && st == 1) || (kf == 3 && kt == 3 && st == 1) || (kf == 8 && kt == 2 && st == 1) ||
                       (kf == 8 && kt == 4 && st == 2);
    const bool chans = (cpg == 2 && opg == 2) || (cpg == 4 && opg == 2) || (cpg == 2 && opg == 4);
    return shape && chans;
}
// output frequency rows per lane: 2 for the 3 x 3 layers (4 input rows instead of 6: 6.9 -> 5.6 ms on freqmpgr1); the strided 8-row layers
// measured slower with 2 (12 rows x two sources in flight: 4.4 -> 5.6 ms), 1 x 1 layers share nothing
static int gconv2d_fo(int kf, int Fo) {
    static const int fo3 = ab_knob("FC_GCONV_FO3", 2);     // A / B aid: 1 .. 4 rows per lane for the 3 x 3 layers
    if (kf != 3 || Fo <= 1) return 1;
    const int f = fo3 < 1 ? 1 : (fo3 > 4 ? 4 : fo3);
    return f < Fo ? f : Fo;
}
bool gconv2d_fuses_halo(int kf, int kt, int st, int Fo, int halo) { return halo > 0 && Fo > halo && !gconv2d_lds3(kf, kt, st); }
int gconv2d_nblk(int Tout, int Fo, int G, int kf) {
#ifdef FC_AB_KNOBS
    if (gconv2d_lds3(kf, kf, 1)) return cdiv(Tout, G3_TN) * cdiv(Fo, G3_RF) * G;       // (the grouped layers are square: kt == kf for kf == 3)
#endif
    return cdiv(Tout, 1024) * cdiv(Fo, gconv2d_fo(kf, Fo)) * G;
}

hipError_t launch_gconv2d(const GConvLaunch& c, hipStream_t st) {
    GConvArgs a;
    a.src0 = c.src0; a.aff0 = c.aff0; a.src1 = c.src1; a.aff1 = c.aff1; a.w = c.w; a.bias = c.bias; a.out = c.out; a.partials = c.partials;
    a.C = c.C; a.M = c.M; a.Tin = c.Tin; a.Tout = c.Tout; a.Fo = c.Fo; a.G = c.G; a.sf = c.sf;
    a.padL = c.padL;
    const int maxpad = c.padL > c.padR ? c.padL : c.padR;
    a.Leff = c.Tin > maxpad ? c.Tin : maxpad + 1;
    a.elu = c.elu; a.alpha = c.alpha;
    a.in_sB = c.in_sB; a.in_sF = c.in_sF; a.out_sB = c.out_sB; a.out_sF = c.out_sF;
    a.out_halo = gconv2d_fuses_halo(c.kf, c.kt, c.st, c.Fo, c.out_halo) ? c.out_halo : 0;
    const int fo_n = gconv2d_fo(c.kf, c.Fo);
    if (c.kf >= 4 ? c.sf != c.kf / 2 : c.sf != 1) return hipErrorInvalidValue;          // the kernel's compile-time row stride
#ifdef FC_AB_KNOBS
    if (gconv2d_lds3(c.kf, c.kt, c.st)) {
        const int cpg3 = c.C / c.G, opg3 = c.M / c.G;
        const int FoT = cdiv(c.Fo, G3_RF);
        if ((long long)c.B * FoT > 65535 || c.G > 65535) return hipErrorInvalidValue;
        dim3 grid3(cdiv(c.Tout, G3_TN), c.G, c.B * FoT), block3(256);
        const size_t lds = (size_t)cpg3 * (G3_RF + 2) * G3_PW * sizeof(float);
        const bool dual3 = c.src1 != nullptr;
#define FC_G3(CP, OP)                                                                                                           \
        if (cpg3 == CP && opg3 == OP) {                                                                                         \
            if (dual3) hipLaunchKernelGGL((gconv2d_3x3_lds_kernel<CP, OP, true>), grid3, block3, lds, st, a);                    \
            else hipLaunchKernelGGL((gconv2d_3x3_lds_kernel<CP, OP, false>), grid3, block3, lds, st, a);                         \
            return hipGetLastError();                                                                                           \
        }
        FC_G3(2, 2) FC_G3(4, 2) FC_G3(2, 4)
#undef FC_G3
        return hipErrorInvalidValue;
    }
#endif
    if ((long long)c.B * cdiv(c.Fo, fo_n) > 65535 || c.G > 65535) return hipErrorInvalidValue;
    dim3 grid(cdiv(c.Tout, 1024), c.G, c.B * cdiv(c.Fo, fo_n)), block(256);
    const int cpg = c.C / c.G, opg = c.M / c.G;
    const bool dual = c.src1 != nullptr;
    // rows not longer than the padding: columns without a source must read as zeros.  The same (gather) instantiation also takes every
    // padding split the register-reflection form of the other one is not written for
    const bool pad_fast = (c.kt == 3 && c.padL + c.padR == 2 && (c.padL == 1 || c.padL == 2)) || (c.kt == 2 && c.padL == 1 && c.padR == 0) || c.kt == 1 ||
                          c.st != 1;
    // unclamped row-end loads only inside a range the caller vouches for (the contract used to be a convention)
    auto guarded = [&](const float* src) {
        if (!src) return true;
        if (!c.guard_lo || !c.guard_hi) return false;
        const char* lo = (const char*)src;
        const char* hi = lo + ((size_t)c.B * (size_t)c.in_sB) * sizeof(float);
        return lo - 16 >= c.guard_lo && hi + 64 <= c.guard_hi;
    };
    const bool nm = a.Leff != a.Tin || !pad_fast || !guarded(c.src0) || !guarded(c.src1);
#define FC_GCL(CP, OP, KF_, KT_, ST_, FO_)                                                                                    \
    do {                                                                                                                     \
        if (dual && nm) hipLaunchKernelGGL((gconv2d_kernel<CP, OP, KF_, KT_, ST_, true, FO_, true>), grid, block, 0, st, a);         \
        else if (dual) hipLaunchKernelGGL((gconv2d_kernel<CP, OP, KF_, KT_, ST_, true, FO_, false>), grid, block, 0, st, a);         \
        else if (nm) hipLaunchKernelGGL((gconv2d_kernel<CP, OP, KF_, KT_, ST_, false, FO_, true>), grid, block, 0, st, a);           \
        else hipLaunchKernelGGL((gconv2d_kernel<CP, OP, KF_, KT_, ST_, false, FO_, false>), grid, block, 0, st, a);                  \
    } while (0)
#ifdef FC_AB_KNOBS      // 3 / 4 output rows per lane (FC_GCONV_FO3): A / B builds only; the shipped library instantiates the tested 1 and 2
#define FC_GC_AB(CP, OP, KF_, KT_, ST_)                                                                                      \
        else if (fo_n == 3) FC_GCL(CP, OP, KF_, KT_, ST_, (KF_ == 3 ? 3 : 1));                                               \
        else if (fo_n == 4) FC_GCL(CP, OP, KF_, KT_, ST_, (KF_ == 3 ? 4 : 1));
#else
#define FC_GC_AB(CP, OP, KF_, KT_, ST_)
#endif
#define FC_GC(CP, OP, KF_, KT_, ST_)                                                                                         \
    if (cpg == CP && opg == OP && c.kf == KF_ && c.kt == KT_ && c.st == ST_) {                                               \
        if (fo_n == 2) FC_GCL(CP, OP, KF_, KT_, ST_, (KF_ == 3 ? 2 : 1));                                                    \
        FC_GC_AB(CP, OP, KF_, KT_, ST_)                                                                                      \
        else FC_GCL(CP, OP, KF_, KT_, ST_, 1);                                                                               \
        return hipGetLastError();                                                                                            \
    }
#define FC_GCS(KF_, KT_, ST_) FC_GC(2, 2, KF_, KT_, ST_) FC_GC(4, 2, KF_, KT_, ST_) FC_GC(2, 4, KF_, KT_, ST_)
    FC_GCS(1, 1, 1) FC_GCS(3, 3, 1) FC_GCS(8, 2, 1) FC_GCS(8, 4, 2)
#undef FC_GCS
#undef FC_GC
#undef FC_GC_AB
#undef FC_GCL
    return hipErrorInvalidValue;
}

// =================================================================================================
// Grouped ConvTranspose2d((2 fr, 2 tr), stride (fr, tr)) with 2 input channels and 1 output channel per group (tr_conv_group_ratio = 1,
// seanet_decoder.py:324): 8 multiply-adds per output, the output tensor is fr * tr / 2 times the input -- a pure store-bound layer.
// Output-centric direct form over the UNTRIMMED output (GroupNorm statistics cover it, conv.py:430-447): row fu = q fr + p reads input rows
// q (tap p) and q - 1 (tap p + fr) of the materialised ELU'd input z (one zero row above and below), column tu = i tr + ph reads
// columns i (tap ph) and i - 1 (tap ph + tr); a lane owns 4 consecutive columns, one workgroup = (utterance, untrimmed row, 1024
// columns) x all output channels in turn (the input rows stay in L1 / L2 across them).  Stores only inside the trimmed window.
// =================================================================================================
struct GConvTrArgs {
    const float* z;          // [B][Fin + 2][C][Tin], rows -1 and Fin zero
    const float *w, *bias;   // torch layout [C][1][2 fr][2 tr], [cout]
    float* out;              // [B][Fout + 2ho][cout][Tout], pointing at row ho
    double* partials;        // [B][(Fin + 1) fr][ttiles][2]
    int C, cout, Fin, Tin, fr, f_l, Fout, trimL, Tout;
    long long out_sB;
    int nx, ny, nz;          // logical grid: column tiles, (input-row pairs x channel chunks) or untrimmed output rows, utterances
    int cs;                  // channel chunks per input-row pair (<= fr)
    int out_halo;            // > 0: reflected halo rows of the (trimmed) output written by the rows' own workgroups (Fout > out_halo)
};

#ifndef FC_GCONVTR_DPP
#define FC_GCONVTR_DPP 1
#endif
#ifndef FC_GCONVTR_XCD
#define FC_GCONVTR_XCD 1
#endif
#ifndef FC_GCONVTR_ROWS
#define FC_GCONVTR_ROWS 1
#endif
template <int TR>
__global__ __launch_bounds__(256) void gconvtr2d_kernel(const GConvTrArgs p) {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    constexpr int NI = 4 / TR + 1;                   // input columns behind 4 untrimmed output columns
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // XCD-aware order (round 4).  Neighbouring input-row pairs share a row (row q is read again for pair q + 1): in dispatch order
    // (workgroup w -> XCD w % 8, observed) they sat on different XCDs and every XCD's L2 fetched the input again.  1-D grid of 8 * per
    // workgroups; XCD x walks the contiguous range [x * per, (x + 1) * per) of the (tile, row, utterance) order.
    int tile, yi, b;
    if (FC_GCONVTR_XCD) {
        const unsigned w = blockIdx.x, per = gridDim.x >> 3;
        const unsigned v = (w & 7u) * per + (w >> 3);
        if (v >= (unsigned)(p.nx * p.ny * p.nz)) return;
        tile = v % p.nx; yi = (v / p.nx) % p.ny; b = v / (p.nx * p.ny);
    } else { tile = blockIdx.x; yi = blockIdx.y; b = blockIdx.z; }
    // FC_GCONVTR_ROWS (round 4): one workgroup = one input-row pair (q, q - 1) and ALL fr output rows q fr .. q fr + fr - 1 it feeds: the four
    // (channel, row) pieces of a channel pair are loaded once and multiplied fr times (fr = 4: 4 load instructions per 4 stores instead of
    // 16 -- the kernel is bound by the address path).  Otherwise one workgroup per untrimmed output row.
    // ... and one of `cs` chunks of the output channels (cs <= fr): the layers behind the bottleneck have few rows and many channels -- 544
    // workgroups x 64 serial channel iterations on 1 536 workgroup slots before the split.
    const int q = FC_GCONVTR_ROWS ? yi / p.cs : yi / p.fr;
    const int chunk = FC_GCONVTR_ROWS ? yi - q * p.cs : 0;
    const int ph_lo = FC_GCONVTR_ROWS ? 0 : yi - q * p.fr, ph_hi = FC_GCONVTR_ROWS ? p.fr : ph_lo + 1;
    const int co_per = (p.cout + p.cs - 1) / p.cs;
    const int co_lo = FC_GCONVTR_ROWS ? chunk * co_per : 0;
    const int co_hi = FC_GCONVTR_ROWS ? (co_lo + co_per < p.cout ? co_lo + co_per : p.cout) : p.cout;
    const int tu0 = tile * 1024 + 4 * tid;           // first untrimmed output column of this lane (a multiple of 4, hence of TR)
    const int Tu = (p.Tin + 1) * TR;
    const bool live = tu0 < Tu;
    const int i0 = tu0 / TR;                         // input column of tap s = 0 for the lane's first output
    const size_t rowsz = (size_t)p.C * p.Tin;
    const float* zq = p.z + ((size_t)b * (p.Fin + 2) + (q + 1)) * rowsz;       // input row q (row Fin = the zero halo)
    const float* zm = zq - rowsz;                                               // input row q - 1 (row -1 = the zero halo)
    const int kt = 2 * TR, kf = 2 * p.fr;
    float s1v = 0.f, s2v = 0.f;
    // wave-uniform: every lane's columns i0 - 1 .. i0 + max(NI, 4) - 2 lie inside the input row (all but the first and the last wave of a row)
    const bool fast = __all(live && i0 >= 1 && i0 + (NI > 4 ? NI - 2 : 2) < p.Tin);
    if (live) {
        for (int co = co_lo; co < co_hi; ++co) {
            float x[2][2][NI];                       // [ci][row q / q - 1][columns i0 - 1 .. i0 + NI - 2]
            if (fast && TR == 1 && FC_GCONVTR_DPP) {
                // TR = 1: the lane's own 4 columns i0 .. i0 + 3 in ONE 16-byte load; column i0 - 1 is the left neighbour's last column
                // (DPP wave_shr:1), only lane 0 of a wave loads it.
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const float* row = (r ? zm : zq) + (size_t)(2 * co + ci) * p.Tin + i0;
                        const f32x4 v = *(const f32x4u*)row;
                        float left = 0.f;
                        if (lane == 0) left = row[-1];
                        left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, left), __builtin_bit_cast(int, v[3]),
                                                                                     0x138, 0xF, 0xF, false));       // wave_shr:1
                        x[ci][r][0] = left;
#pragma unroll
                        for (int j = 1; j < NI; ++j) x[ci][r][j] = v[(j - 1) & 3];
                    }
            } else if (fast) {
                // whole wave inside the row: ONE 16-byte load (+ one dword for the fifth column of TR = 1) per (channel, row) instead of
                // NI dword loads -- the kernel is bound by the address path (PMC: TA busy 89 - 92 %, tools/pmc_ta.sh), not by HBM or latency
                // (issuing the loads of channel pair co + 1 ahead of the arithmetic of pair co changed nothing: 1.64 vs 1.63 ms)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const float* row = (r ? zm : zq) + (size_t)(2 * co + ci) * p.Tin + (i0 - 1);
                        const f32x4 v = *(const f32x4u*)row;
#pragma unroll
                        for (int j = 0; j < (NI < 4 ? NI : 4); ++j) x[ci][r][j] = v[j];
                        if (NI > 4) x[ci][r][NI - 1] = row[4];
                    }
            } else {
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const float* row = (r ? zm : zq) + (size_t)(2 * co + ci) * p.Tin;
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const int t = i0 - 1 + j;
                            x[ci][r][j] = (t >= 0 && t < p.Tin) ? row[t] : 0.f;
                        }
                    }
            }
            const float bm = p.bias[co];
            for (int ph_f = ph_lo; ph_f < ph_hi; ++ph_f) {
                const int fu = q * p.fr + ph_f, fo = fu - p.f_l;
                const bool row_ok = fo >= 0 && fo < p.Fout;
                float acc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = 0.f;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const float* wr = p.w + ((size_t)(2 * co + ci) * kf + ph_f + r * p.fr) * kt;     // uniform: scalar loads
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int ph = j % TR, ii = j / TR;          // output column tu0 + j = (i0 + ii) TR + ph
                            acc[j] = fmaf(wr[ph], x[ci][r][ii + 1], acc[j]);           // tap s = 0: column i0 + ii
                            acc[j] = fmaf(wr[ph + TR], x[ci][r][ii], acc[j]);          // tap s = 1: column i0 + ii - 1
                        }
                    }
                float ov[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ov[j] = acc[j] + bm;
                    if (tu0 + j < Tu) { s1v += ov[j]; s2v = fmaf(ov[j], ov[j], s2v); }
                }
                if (row_ok) {
                    float* orow = p.out + (size_t)b * p.out_sB + ((size_t)fo * p.cout + co) * p.Tout;
                    const int to0 = tu0 - p.trimL;
                    const int h = p.out_halo;
                    const long long rsz = (long long)p.cout * p.Tout;
                    const long long mir0 = (h && fo >= 1 && fo <= h) ? -2ll * fo * rsz : 0;
                    const long long mir1 = (h && fo <= p.Fout - 2 && fo >= p.Fout - 1 - h) ? 2ll * (p.Fout - 1 - fo) * rsz : 0;
                    if (to0 >= 0 && to0 + 3 < p.Tout) {                                                                   // one 16-byte store (round 3)
                        const f32x4 o4 = {ov[0], ov[1], ov[2], ov[3]};
                        *(f32x4u*)(orow + to0) = o4;
                        if (mir0) *(f32x4u*)(orow + mir0 + to0) = o4;
                        if (mir1) *(f32x4u*)(orow + mir1 + to0) = o4;
                    } else
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int to = to0 + j;
                            if (to >= 0 && to < p.Tout) {
                                orow[to] = ov[j];
                                if (mir0) orow[mir0 + to] = ov[j];
                                if (mir1) orow[mir1 + to] = ov[j];
                            }
                        }
                }
            }
        }
    }
    double d1 = (double)s1v, d2 = (double)s2v;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        d1 += __shfl_xor(d1, off, 64);
        d2 += __shfl_xor(d2, off, 64);
    }
    if (lane == 0) { red[0][wid] = d1; red[1][wid] = d2; }
    __syncthreads();
    if (tid == 0 && p.partials) {                        // weight_norm nets: no GroupNorm statistics
        // partial slots stay one per (utterance, untrimmed output row, tile): the cs workgroups of an input-row pair share its fr slots --
        // chunk c fills slot c and zeroes slots c + cs, c + 2 cs, ...
        const int nrows = (p.Fin + 1) * p.fr;
        const int step = FC_GCONVTR_ROWS ? p.cs : 1, first = FC_GCONVTR_ROWS ? chunk : ph_lo;
        for (int ph_f = first; ph_f < ph_hi; ph_f += step) {
            const size_t slot = (((size_t)b * nrows + (q * p.fr + ph_f)) * p.nx + tile) * 2;
            p.partials[slot] = ph_f == first ? ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3] : 0.0;
            p.partials[slot + 1] = ph_f == first ? ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3] : 0.0;
        }
    }
}

bool gconvtr2d_ok(int cpg, int opg, int tr) { return cpg == 2 && opg == 1 && (tr == 1 || tr == 2); }
int gconvtr2d_nblk(int Tin, int tr, int Fin, int fr) { return cdiv((Tin + 1) * tr, 1024) * (Fin + 1) * fr; }

hipError_t launch_gconvtr2d(const float* z, const float* w, const float* bias, float* out, double* partials, int B, int C, int cout, int Fin,
                            int Tin, int fr, int tr, int f_l, int Fout, int trimL, int Tout, long long out_sB, hipStream_t st, int out_halo) {
    GConvTrArgs a;
    a.z = z; a.w = w; a.bias = bias; a.out = out; a.partials = partials;
    a.C = C; a.cout = cout; a.Fin = Fin; a.Tin = Tin; a.fr = fr; a.f_l = f_l; a.Fout = Fout; a.trimL = trimL; a.Tout = Tout; a.out_sB = out_sB;
    a.out_halo = (out_halo > 0 && Fout > out_halo) ? out_halo : 0;
    if ((Fin + 1) * fr > 65535 || B > 65535) return hipErrorInvalidValue;
    static const int cs_env = ab_knob("FC_GCONVTR_CS", 0);       // A / B aid
    a.nx = cdiv((Tin + 1) * tr, 1024);
    a.cs = 1;
    while (a.cs * 2 <= fr && a.cs * 2 <= cout && (long long)a.nx * (Fin + 1) * B * a.cs < 6144) a.cs *= 2;   // >= 4 workgroups per slot, or all there are
    if (cs_env > 0) a.cs = cs_env < fr ? (cs_env < cout ? cs_env : cout) : (fr < cout ? fr : cout);
    a.ny = FC_GCONVTR_ROWS ? (Fin + 1) * a.cs : (Fin + 1) * fr; a.nz = B;
    const long long total = (long long)a.nx * a.ny * a.nz;
    if (total > (1ll << 30)) return hipErrorInvalidValue;
    dim3 grid(a.nx, a.ny, a.nz), block(256);
    if (FC_GCONVTR_XCD) grid = dim3((unsigned)(8 * ((total + 7) / 8)), 1, 1);
    if (tr == 1) hipLaunchKernelGGL(gconvtr2d_kernel<1>, grid, block, 0, st, a);
    else if (tr == 2) hipLaunchKernelGGL(gconvtr2d_kernel<2>, grid, block, 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace fc
