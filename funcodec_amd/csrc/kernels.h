// Launch interface between the engine (engine.hip) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

namespace fc {

// Tuning / ablation switches (A / B aids of tools/*.sh) are read from the environment only in builds made with
// FC_BUILD_DEFINES=FC_AB_KNOBS (and in FC_TIMELINE builds); the shipped library takes the measured-best default of each, so that no
// untested kernel variant can be selected in production.
inline int ab_knob(const char* name, int dflt) {
#ifdef FC_AB_KNOBS
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    // a tuning variable set against the shipped library is IGNORED, and says so once (A / B scripts run against the default
    // build compared identical configurations silently)
    if (getenv(name)) fprintf(stderr, "funcodec_amd: %s is set but ignored by this build (tuning knobs need FC_BUILD_DEFINES=FC_AB_KNOBS)\n", name);
    return dflt;
#endif
}

// Deployment switches: the ONLY environment variables the shipped library reads besides the test hooks below.  Every value of each selects a
// code path whose results are tested against the default's (tests/test_gpu_parity.py::test_staging_scheme_and_workgroup_count_do_not_change_results,
// ::test_lstm_launch_wavefront_fallback_matches_persistent_kernel; tests/test_laura.py persistent == chain), so none can select an untested variant:
//   FC_LSTM_PERSIST=0   per-step LSTM launches instead of the persistent recurrence (shared-GPU deployments: no co-residency requirement)
//   FC_ROW=0            element staging for the stride-1 conv layers too
//   FC_TARGET_WGS=<n>   workgroups a conv launch aims for (default 2 per CU)
//   FC_LAURA_PERSIST=0 / FC_LAURA_GRAPH=0   LauraTTS decoding step as the kernel chain / without the HIP graph (laura.hip)
// Test hooks (change behaviour on purpose, documented where they are read): FC_ABLATE_LSTM=64, FC_LAURA_PERSIST_TEST=timeout.
// Diagnostics (print / dump only): FC_DUMP_PLAN, FC_LAURA_TRACE.
inline int deploy_switch(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// One input of a fused prologue:  v = src[b][c][t];  optional /div[b];  optional per-(b,c) affine
// (GroupNorm apply: v*aff[b][c][0] + aff[b][c][1]).
struct Src {
    const float* ptr = nullptr;   // [B][C][T]
    const float* aff = nullptr;   // [B][C][2] or null
    const float* div = nullptr;   // [B] or null
    int used = 0;                 // host-side bookkeeping only (dry-run planning has null pointers)
};

struct ConvLaunch {
    Src s0, s1;                   // s1.ptr == null -> single source; else v = f0(s0) + f1(s1)
    int elu = 0; float alpha = 1.f;
    const float* wt = nullptr;    // packed weights [mtile][chunk][Wbuf]: [kk][cl][BM] + zero pad to 4 KiB
    const float* bias = nullptr;  // [Mpad]
    const int* koff = nullptr;    // device table from conv_koff_table()
    float* out = nullptr;
    long long out_sB = 0, out_sM = 0, out_sT = 1;
    int B = 0, Cin = 0, Tin = 0;
    int M = 0;                    // real GEMM rows (Cout, or Cout*r for transposed conv)
    int Tout = 0;                 // GEMM columns per utterance
    int k = 1, stride = 1, padL = 0, padR = 0;   // padR includes the reference's "extra" padding
    int dil = 1;                  // tap spacing (dilated residual convs; 1 elsewhere)
    int pad_zero = 0;             // 0 reflect (pad1d), 1 zeros
    int up_r = 0, trimL = 0, Tfinal = 0;          // transposed-conv scatter epilogue when up_r > 0
    double* partials = nullptr;   // [B][nblk][2] (sum, sumsq) or null
    int BM = 128, BN = 128, CC = 2, nchunk = 1;   // tiling chosen at pack time
    int row = 0;                  // stride-1 row staging (conv_row_ok() at pack time)
    int xq_Tp = 0;                // > 0: s0.ptr is the channel-quad-interleaved padded materialisation of launch_combine_xq() with this many
                                  // columns per row (padL + Tin + padR): the slab is staged by DMA (conv_kernel.h MODE 5); no s1, no affine, no ELU
    // 2-D nets in frequency-major layout (kernels of conv_kernel.h, "two-level batch addressing"); defaults = 1-D layer
    int Fo = 1;                   // virtual utterances per real utterance (B = real utterances x Fo)
    int affC = 0;                 // channels of the affine tables (0: = Cin)
    long long in_sB0 = 0, in_sB1 = 0;     // 0: in_sB0 = Cin * Tin
    long long out_sF = 0, part_sB0 = 0;   // 0: part_sB0 = Fo * conv_nblk()
    int store_lo = 0, store_hi = 1 << 30;
    const float* w_plain = nullptr;   // [Cin][k] device copy for single-output-channel layers (FMA kernel), else null
    float bias_host0 = 0.f;           // bias[0] of such a layer
};

int conv_nblk(const ConvLaunch& c);                         // stat partials per utterance
bool conv_cout1_ok(const ConvLaunch& c);                    // launch_conv() will take a few-output FMA kernel (1..4 output channels)
bool conv_fewout_rows(const ConvLaunch& c);                 // ... its LDS form for short rows (else the streaming form)
void conv_fewout_name(const ConvLaunch& c, char* buf, size_t n); // the kernel symbol launch_conv picks for a few-output layer (profile class name)
size_t conv_lds_bytes(const ConvLaunch& c);
int conv_wbuf_floats(int k, int CC, int BM);                // floats per packed weight chunk (4 KiB multiple)
bool conv_quad(int CC);                                     // this chunking uses the quad-k operand layout (kernels.hip)
size_t conv_pack_index(int k, int CC, int BM, int kk, int cl, int mm);   // float index of W[row mm][chunk channel cl][tap kk] in a chunk image
size_t conv_lds_bytes_for(int k, int stride, int dil, int CC, int BM, int BN, int Cin, int ntab, int row, int dma_rounds = -1);
bool conv_row_ok(int k, int stride, int dil, int CC, int BM, int BN, int Cin, bool dual);   // row staging usable with this chunking?
bool conv_slab_fits(int k, int stride, int dil, int CC, int BN, int BM, bool dual);
int conv_wgs_per_cu(int BM);
std::vector<int> conv_koff_table(int k, int stride, int dil, int CC, int BN, int row);
hipError_t launch_conv(const ConvLaunch& c, hipStream_t st);
void conv_variant(const ConvLaunch& c, int* mode, int* nu, int* row);    // template instantiation launch_conv() picks

// Fused head of a thin residual block (kernels.hip 1c): sc = shortcut(x) (k = 1, C -> C) and b1 = block.1(ELU(x)) (k, dilation,
// C -> C/2, reflect padded) from ONE staging of x = f0(s0) [+ f1(s1)].
struct ResHeadLaunch {
    Src s0, s1;
    const float *wsc = nullptr, *wb1 = nullptr;     // LDS images: wsc[c][m] (C x C); wb1[kk*C + c][h] (k*C x C/2)
    const float *bsc = nullptr, *bb1 = nullptr;     // biases [C], [C/2]
    float *out_sc = nullptr, *out_b1 = nullptr;     // [B][C][T], [B][C/2][T] raw conv outputs
    double *part_sc = nullptr, *part_b1 = nullptr;  // [B][reshead_ntiles(T)][2] or null
    int B = 0, C = 0, T = 0, k = 3, dil = 1, padL = 0, padR = 0;
    float alpha = 1.f;
};
bool reshead_ok(int C, int hid, int k_sc, int k_b1, int dil, int stride);
int reshead_ntiles(int T);
hipError_t launch_reshead(const ResHeadLaunch& c, hipStream_t st);

// ---- STFT-domain codec (freq_kernels.hip): frequency-major 2-D activations [B][F + 2*halo][C][T]
// grouped Conv2d with 2 / 4 channels per group (freq_kernels.hip): direct FMA kernel instead of a block-diagonal dense GEMM
struct GConvLaunch {
    const float *src0 = nullptr, *aff0 = nullptr, *src1 = nullptr, *aff1 = nullptr;   // sources at the first row the fo = 0 window reads
    const float *w = nullptr, *bias = nullptr;                                         // [M][C / G][kf][kt] (torch), [M]
    float* out = nullptr;                                                              // at frequency row `halo`
    double* partials = nullptr;                                                        // [B][gconv2d_nblk()][2]
    int B = 0, C = 0, M = 0, G = 1, Tin = 0, Tout = 0, Fo = 0, kf = 1, kt = 1, sf = 1, st = 1, padL = 0, padR = 0, elu = 0;
    float alpha = 1.f;
    long long in_sB = 0, in_sF = 0, out_sB = 0, out_sF = 0;
    // The register-reflection (FASTEDGE) instantiations load every lane's window where it lies: up to 8 bytes in front of a source row and 16
    // bytes past its end.  Only a caller that GUARANTEES readable memory there may opt in: [guard_lo, guard_hi) is the readable byte range both
    // sources lie in (the engine: its workspace, whose first buffer starts 256 bytes in and which ends with 4 KiB of slack, enforced by
    // Ctx::alloc).  Null (the default): every source column is clamped / gathered (the NEEDMASK instantiation), no out-of-row read.
    const char *guard_lo = nullptr, *guard_hi = nullptr;
    int out_halo = 0;      // > 0: the kernel also writes the reflected halo rows of its output (row -i = row i, row Fo-1+i = row Fo-1-i) --
                           // only honoured when gconv2d_fuses_halo() says so, else the caller runs launch_halo_rows
};
bool gconv2d_ok(int cpg, int opg, int kf, int kt, int st);
bool gconv2d_fuses_halo(int kf, int kt, int st, int Fo, int halo);
int gconv2d_nblk(int Tout, int Fo, int G, int kf);
hipError_t launch_gconv2d(const GConvLaunch& c, hipStream_t st);
// grouped ConvTranspose2d((2 fr, 2 tr), stride (fr, tr)), 2 input / 1 output channel per group, over the materialised ELU'd input z
bool gconvtr2d_ok(int cpg, int opg, int tr);
int gconvtr2d_nblk(int Tin, int tr, int Fin, int fr);
hipError_t launch_gconvtr2d(const float* z, const float* w, const float* bias, float* out, double* partials, int B, int C, int cout, int Fin,
                            int Tin, int fr, int tr, int f_l, int Fout, int trimL, int Tout, long long out_sB, hipStream_t st,
                            int out_halo = 0 /* > 0 (and < Fout): reflected halo rows of the output written by the kernel */);
hipError_t launch_polyphase_in(const float* wav, const float* div, int B, int T, int hop, int n_fft, int Mp, float* xp, hipStream_t st);
hipError_t launch_stft_feats(const float* spec, int B, int F, int Tp, long long spec_sB, int halo, int C, float* feats, hipStream_t st);
hipError_t launch_feats_relayout(float* eng, float* ref, int B, int C, int F, int Tp, int halo, int to_ref, hipStream_t st);
hipError_t launch_halo_rows(float* buf, int B, int F, int halo, int C, int T, int zero, hipStream_t st);
hipError_t launch_combine2d(const float* s0, const float* aff0, int h0, const float* s1, const float* aff1, int h1, int elu, float alpha,
                            int B, int F, int C, int T, float* dst, int hd, hipStream_t st,
                            int halo_mode = 0 /* 1: also write dst's hd reflected halo rows (needs F > hd), 2: zero them */);
hipError_t launch_spec_from_dec(const float* dec, const float* aff, int B, int F, int Tp, int halo, int C, float* spec, hipStream_t st);
hipError_t launch_istft_finish(const float* ypoly, const float* win2, int B, int hop, int n_fft, int Mp, int Tp, const float* mul, int out_len,
                               float* wav, hipStream_t st);

// Reduce stat partials -> mean/rstd -> per-(b,c) GroupNorm affine table aff[b][c] = (rstd*gamma, beta-mean*rstd*gamma)
hipError_t launch_gn_finalize(const double* partials, int nblk, double count, const float* gamma,
                              const float* beta, int C, float eps, int B, float* aff, hipStream_t st);

// xq[b][c / 4][padL + t][c % 4] = pad( [elu]( f0(s0) + f1(s1) ) )[t]  for t in [-padL, Tin + padR): the activated input of a conv layer
// materialised WITH its padding (reflect as pad1d conv.py:82-99 incl. the zero-extension of short inputs, or zeros) and with 4 channels
// interleaved, i.e. in the order the quad-layout conv kernel's B-operand planes hold it: a plane row is contiguous memory and the conv
// kernel stages it by DMA.  C % 4 == 0.  conv_xq_ok(): this layer / launch can take that path; conv_xq_floats(): buffer size incl. the
// slack the last tile's DMA reads past the last row.
bool conv_xq_ok(int Cin, int CC, int k, int stride, int dil, int BM, int BN, int row);
size_t conv_xq_floats(int B, int Cin, int Tp, int BN, int stride, int k, int dil);
hipError_t launch_combine_xq(const Src& s0, const Src& s1, int elu, float alpha, int B, int C, int Tin, int padL, int padR, int pad_zero,
                             float* xq, hipStream_t st);

// out[b][c][t] (strides) = [elu]( f0(s0) + f1(s1) ) * mul[b]     for t < Tcopy
hipError_t launch_combine(const Src& s0, const Src& s1, int elu, float alpha, const float* mul,
                          int B, int C, int Tsrc, int Tcopy, float* out,
                          long long o_sB, long long o_sC, long long o_sT, hipStream_t st);

// x = tanh(x) * range in place over n floats (CostumeQuantizer.input_act, costume_quantizer.py:32-35,66-67)
hipError_t launch_tanh_range(float* x, size_t n, float range, hipStream_t st);

// scale[b] = 1e-8 + sqrt(mean_t x[b][t]^2)
hipError_t launch_volume(const float* wav /* [B][C][T], C = 1 | 2 */, int B, int C, int T, float* scale, hipStream_t st);

// [B][T][D] -> [B][D][T]
hipError_t launch_transpose_btd(const float* in, int B, int T, int D, float* out, hipStream_t st);

// Residual vector quantiser, all stages fused.  x [N][D] rows; cb [nq][K][D]; enorm [nq][K].
hipError_t launch_rvq_encode(const float* x, int N, int D, int K, int nq, const float* cb, const float* cb_frag /* or null */,
                             const float* enorm,
                             int64_t* codes /*[nq][N]*/, float* quant /*[N][D] or null*/,
                             float* quant_bdt /*[B][D][Tf] or null*/, float* subq /*[nq][B][D][Tf] or null*/,
                             int Tf, hipStream_t st, const int* src0 /* [N] stage-0 source rows (q0_ds_ratio > 1) or null */ = nullptr);

// quantizer_conf.q0_ds_ratio > 1 (DistributedResidualVectorQuantization.forward, ddp_core_vq.py:396-404): the first stage quantises
// F.interpolate(residual, size=[Tf // 2]) and its output (and indices) go back through F.interpolate(size=[Tf]), both mode "nearest"
// -- whatever the ratio's value is, the reference halves.  Row-wise that is: stage 0 of frame t quantises frame q0_source_frame(t, Tf).
// torch's nearest source index (ATen UpSample.h nearest_idx / UpSampleKernel.cpp HelperInterpNearest): min(int(floorf(dst * scale)),
// in - 1) with scale = float(in) / out computed in fp32; out == 2 * in reduces to dst >> 1.  tests/test_host.py checks this function
// against F.interpolate for every Tf in 2 .. 4096.
__host__ __device__ inline int q0_source_frame(int t, int Tf) {
    const int half = Tf / 2;
    int j;                                              // half-rate frame that output frame t copies (up: half -> Tf)
    if (Tf == 2 * half) j = t >> 1;
    else {
        const float up = (float)half / (float)Tf;
        j = (int)floorf((float)t * up);
        if (j > half - 1) j = half - 1;
    }
    const float down = (float)Tf / (float)half;          // input frame that half-rate frame j copied (down: Tf -> half)
    int s = (int)floorf((float)j * down);
    if (s > Tf - 1) s = Tf - 1;
    return s;
}
hipError_t launch_q0_map(int* map /* [B][Tf] */, int B, int Tf, hipStream_t st);
// codes [B][Tf][nq] (i64) -> emb [B][Tf][D] and/or emb_bdt [B][D][Tf]
// status: host-visible engine status words (FC_STATUS_*), or null; an index outside [0, K) sets FC_STATUS_BAD_CODE
hipError_t launch_rvq_decode(const int64_t* codes, int B, int Tf, int nq, int D, int K, const float* cb,
                             float* emb, float* emb_bdt, unsigned* status, hipStream_t st);

// engine status words (host-pinned, device-mapped): written by kernels with plain stores, read by the host without a sync
#define FC_STATUS_LSTM_TIMEOUT 0   // persistent LSTM: the grid barrier timed out (workgroups not co-resident); outputs poisoned
#define FC_STATUS_BAD_CODE 1       // decode: a code index outside [0, codebook_size) was clamped
#define FC_STATUS_WORDS 16

#define FC_LSTM_MAX_LAYERS 4
// Layer-wavefront step s (see kernels.hip): w[0] = W_hh0 perm [4H][H]; w[l>=1] = [W_ih_l | W_hh_l] perm [4H][2H];
// bias[l>=1] = perm(b_ih + b_hh); h [L][2][B][H] and c [L][B][H] zero-initialised by the caller.
hipError_t launch_lstm_wave(const float* const* w, const float* const* bias, const float* xproj, float* h, float* c,
                            float* y, int B, int H, int T, int L, int s, hipStream_t st);

// Persistent 2-layer recurrence (one launch); sync = 2 zeroed words; h [2][2][B][H] zeroed by the caller.
hipError_t launch_lstm_persist(const float* w0, const float* w1, const float* bias1, const float* xproj, float* state, float* y,
                               int B, int H, int T, unsigned* status, hipStream_t st);
// _linear_overlap_add (codec_basic.py:77-116): frames = DEVICE array of n_frames device pointers ([B][lens[f]] each), lens = DEVICE
// array; L0 = length of frame 0 (sizes the triangle window); out [B][out_len] = the first out_len samples of the sum
hipError_t launch_overlap_add(const float* const* frames, const int* lens, int n_frames, int B, int L0, int stride, int out_len,
                              float* out, hipStream_t st);
hipError_t debug_timeline(unsigned long long* dst);      // [2 roles][24 items][8 slots], profiling builds only
hipError_t launch_zero_fill(float* p, size_t n, hipStream_t st);
size_t lstm_persist_state_floats(int B, int H, int T);   // barrier words + zero slot + hidden-state history of both layers
size_t lstm_persist_clear_floats(int B, int H);
bool lstm_persist_supported(int B, int H, int L, int device);

}  // namespace fc
