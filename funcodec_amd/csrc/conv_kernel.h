// The implicit-GEMM conv kernel template and its launch ladder (mode x staging variant).  Included by kernels.hip (which only
// DECLARES the per-tile launchers) and by one conv_tile_*.hip per tile shape (which instantiates them): the ~180 instantiations
// then compile as five parallel translation units instead of one 4-minute one.
#pragma once
#include "kernels.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace fc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline __host__ __device__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

// =================================================================================================
// 1. Implicit-GEMM Conv1d / ConvTranspose1d with fused prologue and GroupNorm-statistics epilogue
//
//    out[b][m][n] = bias[m] + sum_{ci,kk} W[m][ci][kk] * f(in[b][ci][n*stride + kk - padL])
//
//    f = the pending elementwise work of the PRODUCING layers, applied while the input slab is staged
//        into LDS: GroupNorm apply (per-(b,c) scale/shift), residual add of a second tensor, ELU,
//        reflect / zero padding (reference: pad1d conv.py:82-99, SConv1d.forward conv.py:243-261,
//        GroupNorm conv.py:45-52, ELU activations.py:24-30, SEANetResnetBlock.forward
//        seanet_encoder.py:60-61).
//    epilogue: + bias, store the RAW conv output once, and emit deterministic per-workgroup partial
//        (sum, sum of squares) in fp64 for this layer's own GroupNorm(1,C) statistics.
//    ConvTranspose1d(k=2r, stride=r) is the same GEMM with M = Cout*r rows (m = co*r + phase), two taps
//        (x[i-1], x[i]) and a scatter store out[co][i*r + phase - trimL]; its statistics cover the
//        UNTRIMMED output as in SConvTranspose1d.forward (conv.py:287-303).
//
//    LDS:  Ws[2][Kc][BM]   weight chunk, k-major so the A fragment (lane -> row) is conflict free; filled by
//                          global_load_lds DMA, double buffered (chunk c+1 streams in while chunk c computes)
//          Xs[2][CC][S][PL] input slab, double buffered, split by stride phase so the B fragment (lane -> column) is
//                          conflict free for every stride (tau = n*S + kk -> [kk % S][n + kk / S]); stride-1 layers
//                          use rows of BN + k - 1 columns padded to 16 bytes (ROW staging)
//          tab[Cin]        the producers' GroupNorm affine for this utterance
//          kofs, bias, red the k-step -> slab offset table, the tile's bias, per-lane GroupNorm partials of two tiles
// =================================================================================================
struct ConvArgs {
    const float *src0, *aff0, *div0, *src1, *aff1;
    const float *wt, *bias;
    float* out;
    double* partials;
    long long out_sB, out_sM, out_sT;
    int B, Cin, Tin, M, Tout, k, stride, padL, padR, pad_zero, Leff;
    int dil;                // tap spacing: tap kk reads slab column n*stride + kk*dil
    int up_r, trimL, Tfinal;
    unsigned magic_r;       // ceil(2^32 / up_r)
    int elu; float alpha;
    int CC, nchunk, Kc;
    int Wbuf;               // floats per packed weight chunk (multiple of 1024)
    int slabW, PL, rowStride;
    int row;                // stride-1 row staging (16-byte loads / LDS stores, one channel row per 32 or 64 lanes)
    int xsf;                // floats per slab buffer
    unsigned magic_slabW;   // floor(2^32 / slabW) + 1
    const int* koff;        // [koff_n] B-operand LDS float offset per k-step (padded, multiple of 4)
    int koff_n;
    int cin_tail;           // Cin % CC != 0: the last chunk runs past the real channels
    // two-level batch addressing (2-D nets in frequency-major layout [B][F][C][T]: a frequency row of an utterance is one
    // "virtual utterance" of the 1-D conv): blockIdx.z = breal*Fo + fo.  1-D layers: Fo = 1, in_sB0 = Cin*Tin, affC = Cin.
    int Fo;                 // virtual utterances (output frequency rows) per real utterance
    int affC;               // channels of the affine tables (table index = channel % affC)
    long long in_sB0, in_sB1;     // floats between real utterances / consecutive fo of the inputs
    long long out_sF;             // floats between consecutive fo of the output (out_sB: between real utterances)
    long long part_sB0;           // partial (sum, sumsq) pairs between real utterances; fo-th row at fo * nblk
    int store_lo, store_hi;       // only virtual utterances fo in [store_lo, store_hi) store their outputs (all contribute statistics)
    int ablate;             // profiling aid (FC_ABLATE env): 1 no MFMA, 2 no stores, 4 no slab loads, 16 no weight DMA,
                            // 128 no epilogue.  0 in production.
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// ELU(alpha) on the hardware exp2: exp(v) = 2^(v*log2 e).  For v <= 0 the rounding of the product contributes
// |v|*log2(e)*2^-24 relative error to e^v, i.e. at most 3e-8 absolute on the ELU output (below one fp32 ulp of the
// result), so no compensated product is needed.  fp32 MFMA and fp32 VALU share the SIMD's FMA lanes on gfx950
// (tests/micro/mfma_valu_overlap.hip: the two do not overlap), so every VALU instruction here is paid in full.
__device__ __forceinline__ float elu_f(float v, float alpha) {
    const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f);
    return v > 0.f ? v : fmaf(e, alpha, -alpha);
}

// Element staging: NU = register slots (slab elements) per staging thread per chunk, a compile-time constant so that the staging
// code is straight-line (a run-time slot count put a scalar branch between every load: measured 15 % slower).  The strided
// layers' slabs are 4.03 / 8.06 / 10.1 / 16.0 / 16.1 x 256 elements, so the instantiations are 5, 9, 11, 16 and 18 slots (two
// sizes, 8 and 16, ran those layers with up to half of the slots empty, and every staging instruction is paid on top of the MFMA
// time: fp32 MFMA and VALU share the SIMD's lanes).  Two-source prologues hold two values and two table entries per slot: <= 9.
constexpr int NU_BIG = 18;
constexpr int NU_DUAL = 9;
constexpr int SLAB_PER_THREAD = NU_BIG;      // register-staged slab elements per thread per chunk (NU = 8 or NU_BIG)

// Direct global -> LDS copy of one packed weight chunk (contiguous, multiple of 4 KiB): each wave
// instruction moves 1 KiB (64 lanes x 16 B) with no VGPR round trip.
__device__ __forceinline__ void dma_weights(const float* __restrict__ gsrc, float* lds_dst, int nfloats, int tid, int ablate = 0) {
    if (ablate & 16) return;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int off = 0; off < nfloats; off += 1024) {
        const float* g = gsrc + off + tid * 4;
        float* l = lds_dst + off + wid * 256;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
    }
}


// MODE 0: plain single source (already activated input, no prologue math)
// MODE 1/2: single source with GroupNorm affine (optional /div), without / with ELU
// MODE 3/4: two summed sources with GroupNorm affines, without / with ELU
// NU: slab elements staged per staging thread per chunk (compile time)
//
// Workgroup = 8 waves with two ROLES (wave specialisation):
//   waves 0-3 "matrix": LDS fragment reads, MFMAs, epilogue stores, per-lane GroupNorm partials (and the weight DMA of
//                        layers with prologue math);
//   waves 4-7 "staging": global loads of the NEXT slab(s) into registers, prologue math (affine / residual / ELU /
//                        padding), the write into the other half of the double-buffered LDS slab, the weight DMA of PLAIN
//                        layers, and the fixed-order reduction of a finished tile's GroupNorm partials.
// Each SIMD hosts one matrix wave and one staging wave of a workgroup, so prologue VALU work and memory latency
// overlap the matrix pipe by construction.  One barrier per K-chunk.  A workgroup owns one (utterance, M tile) and
// a contiguous range of N tiles; the pipeline runs across chunk and tile boundaries.
#ifdef FC_TIMELINE
// Profiling build only (python -m funcodec_amd.build with FC_TIMELINE=1): wave 0 of each role of workgroup (1, 0, 0)
// stamps s_memtime at its phase boundaries for the first 24 work items; read back with fc_debug_timeline().
static __device__ unsigned long long g_timeline[2][24][8];   // one copy per translation unit (tile file)
#define FC_STAMP(role_, f_, slot_)                                                                          \
    do {                                                                                                    \
        if (bx == 1 && mt == 0 && b == 0 && wid == 0 && lane == 0 && (f_) < 24)                               \
            g_timeline[role_][f_][slot_] = (slot_) == 7 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define FC_STAMP(role_, f_, slot_) do {} while (0)
#endif
template <int BM, int BN, int WM, int WN, int MODE, int NU, bool ROW>
__global__ __launch_bounds__(512, 4) void conv_mfma_kernel(const ConvArgs p) {
    static_assert(WM * WN == 4, "4 matrix waves per workgroup");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr bool PLAIN = MODE == 0;                 // MODE: 0 plain | 1 affine | 2 affine+ELU | 3 dual | 4 dual+ELU
    constexpr bool DUAL = MODE >= 3;
    constexpr bool ELU = MODE == 2 || MODE == 4;
    // which role streams the weight chunks: the staging waves when they have no prologue math (PLAIN), else the matrix
    // waves, one DMA piece per loop trip (measured: each choice loses 5-8 % on the other kind of layer)
#ifdef FC_EXP_MATRIX_DMA
    constexpr bool STAGING_DMA = false;
#else
    constexpr bool STAGING_DMA = PLAIN;
#endif
    constexpr bool DEEP = PLAIN;                      // two register sets of staged input in flight
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int XSF = p.xsf;                            // floats per slab buffer (host: image + pad, see make_args)
    float* Xs0 = smem + 2 * p.Wbuf;                   // slab, double buffered
    const int cin_pad = (p.Cin + 1) & ~1;             // keeps everything behind the tables 16-byte aligned
    float2* tab0 = (float2*)(Xs0 + 2 * XSF);
    float2* tab1 = tab0 + (PLAIN ? 0 : cin_pad);
    int* kofs_i = (int*)(tab1 + (DUAL ? cin_pad : 0));
    float* bias_s = (float*)(kofs_i + p.koff_n);
    float2* red = (float2*)(bias_s + BM);             // [2 tiles in flight][256 matrix lanes] (sum, sum of squares)

    const int tid = threadIdx.x;
    const int role = __builtin_amdgcn_readfirstlane(tid >> 8);    // 0 matrix, 1 staging
    const int rtid = tid & 255, lane = tid & 63, wid = (tid >> 6) & 3;
    const int bx = blockIdx.x, mt = blockIdx.y, b = blockIdx.z;
    const int m0 = mt * BM;
    const int ntiles = (p.Tout + BN - 1) / BN;
    const int t_begin = (int)(((long long)ntiles * bx) / gridDim.x);
    const int t_end = (int)(((long long)ntiles * (bx + 1)) / gridDim.x);
    if (t_begin >= t_end) return;
    const int breal = p.Fo > 1 ? b / p.Fo : b, fo = p.Fo > 1 ? b - breal * p.Fo : 0;
    const size_t in_off = (size_t)breal * (size_t)p.in_sB0 + (size_t)fo * (size_t)p.in_sB1;
    const size_t affbase = (size_t)breal * p.affC;
    const bool store_ok = fo >= p.store_lo && fo < p.store_hi;
    // wave-uniform bases of this (virtual) utterance, computed once
    const size_t out_off0 = (size_t)breal * (size_t)p.out_sB + (size_t)fo * (size_t)p.out_sF;
    const size_t part_off0 = ((size_t)breal * (size_t)p.part_sB0 + (size_t)fo * ((size_t)((p.Tout + BN - 1) / BN) * gridDim.y)) * 2;
    const int nitems = (t_end - t_begin) * p.nchunk;  // flattened (tile, chunk) work items of this workgroup
    const bool resident = p.nchunk <= 2;              // the whole K extent of this M tile stays in LDS
    const float* wt_tile = p.wt + (size_t)mt * p.nchunk * p.Wbuf;

    // ---- common prologue: tables -------------------------------------------------------------------
    if (!PLAIN) {   // per-(b, channel) GroupNorm affine of the producers, staged once per workgroup
        for (int c = tid; c < p.Cin; c += 512) {
            const int ca = p.affC == p.Cin ? c : c % p.affC;
            tab0[c] = p.aff0 ? ((const float2*)p.aff0)[affbase + ca] : make_float2(1.f, 0.f);
            if (DUAL) tab1[c] = p.aff1 ? ((const float2*)p.aff1)[affbase + ca] : make_float2(1.f, 0.f);
        }
    }
    for (int i = tid; i < p.koff_n; i += 512) kofs_i[i] = p.koff[i];
    for (int i = tid; i < BM; i += 512) bias_s[i] = p.bias[m0 + i];
    __syncthreads();

    if (role == 1) {
        // =========================================== staging waves =====================================
        const float* __restrict__ s0b = p.src0 + in_off;              // wave-uniform bases, 32-bit lane offsets
        const float* __restrict__ s1b = DUAL ? p.src1 + in_off : p.src0;
        // GroupNorm partial of a finished tile: fixed-order fp64 reduction of the 256 per-lane fp32 partials the
        // matrix waves left in LDS (done here because the staging waves idle at the barrier anyway)
        auto flush_stats = [&](int tile) __attribute__((always_inline)) {
            if (!p.partials || wid != 0) return;
            const float2* r = red + (tile & 1) * 256;
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float2 v = r[lane + 64 * j]; d1 += (double)v.x; d2 += (double)v.y; }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                d1 += __shfl_xor(d1, o, 64);
                d2 += __shfl_xor(d2, o, 64);
            }
            if (lane == 0) {
                const size_t slot_p = part_off0 + ((size_t)mt * ntiles + tile) * 2;
                p.partials[slot_p] = d1;
                p.partials[slot_p + 1] = d2;
            }
        };
        // PLAIN staging (no prologue math) has time to spare: these waves then also stream the weight chunks (global ->
        // LDS DMA); issuing a DMA piece between MFMAs costs the issuing wave 100+ cycles
        if (STAGING_DMA) {
            dma_weights(wt_tile, smem, p.Wbuf, rtid, p.ablate);
            if (resident && p.nchunk == 2) dma_weights(wt_tile + p.Wbuf, smem + p.Wbuf, p.Wbuf, rtid, p.ablate);
        }
        if constexpr (ROW) {
            // ---------------------------------------------------------------------------------------------
            // Row staging (stride-1 layers).  A slab row = one input channel, BN main columns + k-1 tail columns.
            // BN/4 lanes cover the main columns of a row with ONE 16-byte global load and ONE 16-byte LDS store each
            // (64/(BN/4) rows per wave instruction, the 4 waves take consecutive row groups: 4 or 8 rows per round,
            // CC/rows-per-round rounds per item); the CC*(k-1) tail elements go one per thread.  Per element that is
            // 1/4 load + 1/4 store + the prologue math, against 1 load + 1 address add + 1 table read + 1 store per
            // element of the general path, and the only per-element registers are the values themselves, so the K
            // chunk of an item is bounded by LDS and not by the staging registers.
            // Edge tiles (reflect / zero padding inside the slab) load the 4 columns of a lane with 4 dword loads
            // from per-tile column offsets instead.
            // ---------------------------------------------------------------------------------------------
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            constexpr int LPR = BN / 4, RPI = 64 / LPR, RPR = 4 * RPI;     // lanes per row, rows per instruction / round
            // rounds per item: a compile-time constant in the specialised instantiations (NU = 2 / 4 / 8 here) so that the loads
            // of an item are straight-line code.  With a run-time count every load sat behind its own scalar branch, the loaded
            // registers became loop phis, and hipcc copied them (after an s_waitcnt vmcnt(0)) right behind the load issue: the
            // "prefetch" was waited for before the barrier of the same step.  NU = 1: generic fallback (run-time count).
            constexpr bool RFIX = NU > 1;
            constexpr int MAXR = RFIX ? NU : (DUAL ? 4 : 8);
            const int NR = RFIX ? NU : p.CC / RPR;         // host: 1 <= NR <= MAXR
            const int rsub = wid * RPI + lane / LPR;       // my row inside a round
            const int c4 = lane % LPR;                     // my 4-column group
            const unsigned slot0 = 4u * (unsigned)(rsub * p.rowStride + 4 * c4);
            const unsigned lds_round = 4u * (unsigned)(RPR * p.rowStride);
            const size_t src_round = (size_t)RPR * p.Tin;  // floats between the rows of consecutive rounds
            const int km1 = p.slabW - BN;                  // tail columns of a row: (k - 1) * dilation
            const bool has_tail = rtid < p.CC * km1;       // lanes without a tail element re-read their first main element (unconditional load)
            const int t_cl = has_tail ? rtid / km1 : 0, t_j = has_tail ? rtid - t_cl * km1 : 0;
            const unsigned t_slot = 4u * (unsigned)(t_cl * p.rowStride + BN + t_j);
            // per-tile state of the item being LOADED, and of the registers waiting to be written
            unsigned src_off = 0, eoff[4] = {0, 0, 0, 0}, emask = 0, t_off = 0;
            bool ld_edge = false, t_ok = true;
            bool r_edge = false, r_tok = true; unsigned r_emask = 0;
            f32x4 v0[MAXR], v1[DUAL ? MAXR : 1];
            float tv0 = 0.f, tv1 = 0.f;
            int ld_tile = t_begin, ld_chunk = 0, wr_chunk = 0;
            const int hi_lim = p.Tin + p.padR, refl = 2 * (p.Leff - 1);
            auto resolve = [&](int g, bool& ok) __attribute__((always_inline)) {   // slab column (global time index) -> source index
                ok = g >= -p.padL && g < hi_lim;
                int src = g < 0 ? -g : g;
                src = src >= p.Leff ? refl - src : src;
                if (p.pad_zero) { src = g; ok = ok && g >= 0; }
                ok = ok && src < p.Tin;                  // zero padding / zero-extension of short inputs (conv.py:89-93)
                return ok ? src : 0;
            };
            auto setup_tile = [&](int tbase) __attribute__((always_inline)) {
                ld_edge = !(tbase >= 0 && tbase + p.slabW <= p.Tin);
                if (!ld_edge) {
                    src_off = 4u * (unsigned)(rsub * p.Tin + tbase + 4 * c4);
                    t_off = has_tail ? 4u * (unsigned)(t_cl * p.Tin + tbase + BN + t_j) : src_off;
                    t_ok = true;
                } else {
                    emask = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bool ok;
                        const int src = resolve(tbase + 4 * c4 + j, ok);
                        eoff[j] = 4u * (unsigned)(rsub * p.Tin + src);
                        emask |= (ok ? 1u : 0u) << j;
                    }
                    const int src = resolve(has_tail ? tbase + BN + t_j : tbase + 4 * c4, t_ok);
                    t_off = 4u * (unsigned)((has_tail ? t_cl : rsub) * p.Tin + src);
                }
            };
            auto load_slab = [&]() __attribute__((always_inline)) {
                const int tbase = ld_tile * BN - p.padL;
                if (ld_chunk == 0) setup_tile(tbase);
                const size_t cbase = (size_t)(ld_chunk * p.CC) * p.Tin;
                if (++ld_chunk == p.nchunk) { ld_chunk = 0; ++ld_tile; }
                r_edge = ld_edge; r_emask = emask; r_tok = t_ok;
                if (p.ablate & 4) return;
                const float* r0 = s0b + cbase;
                const float* r1 = s1b + cbase;
                if (!ld_edge) {
#pragma unroll
                    for (int r = 0; r < MAXR; ++r) {
                        if (RFIX || r < NR) {
                            v0[r] = *(const f32x4u*)((const char*)(r0 + r * src_round) + src_off);
                            if (DUAL) v1[r] = *(const f32x4u*)((const char*)(r1 + r * src_round) + src_off);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < MAXR; ++r) {
                        if (RFIX || r < NR) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                v0[r][j] = *(const float*)((const char*)(r0 + r * src_round) + eoff[j]);
                                if (DUAL) v1[r][j] = *(const float*)((const char*)(r1 + r * src_round) + eoff[j]);
                            }
                        }
                    }
                }
                tv0 = *(const float*)((const char*)r0 + t_off);
                if (DUAL) tv1 = *(const float*)((const char*)r1 + t_off);
            };
            auto prologue = [&](float v, float w, float2 a, float2 a1) __attribute__((always_inline)) {
                if (PLAIN) return v;
                v = fmaf(v, a.x, a.y);
                if (DUAL) v = v + fmaf(w, a1.x, a1.y);
                if (ELU) v = elu_f(v, p.alpha);
                return v;
            };
            auto write_slab = [&](char* Xd) __attribute__((always_inline)) {
                const int c0 = wr_chunk * p.CC;
                if (++wr_chunk == p.nchunk) wr_chunk = 0;
                float2 a[PLAIN ? 1 : MAXR], a1[DUAL ? MAXR : 1], ta = {1.f, 0.f}, ta1 = {1.f, 0.f};
                if (!PLAIN) {                             // all table reads ahead of the stores (possible aliasing)
#pragma unroll
                    for (int r = 0; r < MAXR; ++r) {
                        if (RFIX || r < NR) {
                            a[r] = tab0[c0 + r * RPR + rsub];
                            if (DUAL) a1[r] = tab1[c0 + r * RPR + rsub];
                        }
                    }
                    ta = tab0[c0 + t_cl];
                    if (DUAL) ta1 = tab1[c0 + t_cl];
                }
#pragma unroll
                for (int r = 0; r < MAXR; ++r) {
                    if (RFIX || r < NR) {
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = prologue(v0[r][j], DUAL ? v1[r][j] : 0.f, PLAIN ? ta : a[r], DUAL ? a1[r] : ta1);
                            if (r_edge) v[j] = ((r_emask >> j) & 1u) ? v[j] : 0.f;
                        }
                        *(f32x4*)(Xd + slot0 + r * lds_round) = v;
                    }
                }
                if (has_tail) {
                    float v = prologue(tv0, tv1, ta, ta1);
                    v = r_tok ? v : 0.f;
                    *(float*)(Xd + t_slot) = v;
                }
            };
            if (STAGING_DMA) {
                dma_weights(wt_tile, smem, p.Wbuf, rtid, p.ablate);
                if (resident && p.nchunk == 2) dma_weights(wt_tile + p.Wbuf, smem + p.Wbuf, p.Wbuf, rtid, p.ablate);
            }
            load_slab();
            write_slab((char*)Xs0);
            if (nitems > 1) load_slab();
            __syncthreads();                              // B0
            int st_tile = t_begin, st_chunk = 0;
            for (int f = 0; f < nitems; ++f) {
                FC_STAMP(1, f, 0);
                if (STAGING_DMA && f + 1 < nitems && !resident) {
                    const int nc = st_chunk + 1 == p.nchunk ? 0 : st_chunk + 1;
                    dma_weights(wt_tile + (size_t)nc * p.Wbuf, smem + ((f + 1) & 1) * p.Wbuf, p.Wbuf, rtid, p.ablate);
                }
                if (f + 1 < nitems) {
                    write_slab((char*)(Xs0 + ((f + 1) & 1) * XSF));
                    FC_STAMP(1, f, 1);
                    if (f + 2 < nitems) load_slab();
                    FC_STAMP(1, f, 2);
                }
                __syncthreads();                          // B(f+1)
                FC_STAMP(1, f, 3);
                if (++st_chunk == p.nchunk) { st_chunk = 0; flush_stats(st_tile); ++st_tile; }
                FC_STAMP(1, f, 4);
            }
            __syncthreads();                              // final (kept symmetric with the matrix role)
            return;
        }
        const int total = p.CC * p.slabW;             // <= 256 * NU
        const float divv = (MODE == 1 && p.div0) ? p.div0[breal] : 1.f;
        // element e = rtid + 256*u of every chunk of every tile maps to the same (local channel cl, slab column tau):
        //   base0[u] = cl*Tin + tau      slot[u] = LDS float index | cl << 16
        // Per-element descriptors, all in BYTES and unpacked (every extraction / shift would be a VALU instruction per
        // element per chunk, and VALU time adds to MFMA time on this chip):
        //   base0[u] = 4*(cl*Tin + tau)   source byte offset relative to the chunk / tile origin
        //   slot[u]  = byte offset of the element inside a slab buffer (dummy slot for lanes without an element)
        //   cl8[u]   = 8*cl               byte offset into the affine tables
        unsigned base0[NU], slot[NU], cl8[PLAIN ? 1 : NU];
        // TWO register sets: the loads of item f+3 are issued while the values of item f+2 are still in flight, so a load
        // has two item times (not one) to return -- the T=250 layers saw ~2.6 us load latency against ~2.9 us items
        constexpr int NSET = DEEP ? 2 : 1;
        float v0[NSET][NU], v1[NSET][DUAL ? NU : 1];
        unsigned inmask = 0, vmask[NSET] = {};
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int e = rtid + 256 * u;
            base0[u] = 0u; slot[u] = (unsigned)(XSF - 1) * 4u;
            if (!PLAIN) cl8[u] = 0u;
            if (e < total) {
                const int cl = (int)__umulhi((unsigned)e, p.magic_slabW);
                const int tau = e - cl * p.slabW;
                int ph, q;
                switch (p.stride) {
                    case 1: ph = 0; q = tau; break;
                    case 2: q = tau >> 1; ph = tau & 1; break;
                    case 4: q = tau >> 2; ph = tau & 3; break;
                    case 8: q = tau >> 3; ph = tau & 7; break;
                    default: q = tau / p.stride; ph = tau - q * p.stride; break;
                }
                base0[u] = 4u * (unsigned)(cl * p.Tin + tau);
                slot[u] = 4u * (unsigned)(cl * p.rowStride + ph * p.PL + q);
                if (!PLAIN) cl8[u] = (unsigned)cl * 8u;
                inmask |= 1u << u;
            }
        }
        // base0 holds the descriptors of the tile being LOADED: the static ones above for interior tiles, per-tile
        // reflected / zero-padded ones for the (at most two) edge tiles of a row -- computed once per tile, not per
        // chunk: the index math is ~25 VALU instructions per element and VALU time adds to MFMA time on this chip
        bool base_static = true, ld_interior = true;
        unsigned tile_mask = 0;
        bool all_valid[NSET] = {};                    // the set's register contents need no padding mask
        // (tile, chunk) cursors advance incrementally: no integer divisions on the per-chunk path
        int ld_tile = t_begin, ld_chunk = 0, wr_chunk = 0;
        auto setup_tile = [&](int tbase) __attribute__((always_inline)) {
            ld_interior = tbase >= 0 && tbase + p.slabW <= p.Tin;
            tile_mask = inmask;
            if (ld_interior && base_static) return;
            const int hi_lim = p.Tin + p.padR, refl = 2 * (p.Leff - 1);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                unsigned ee = (unsigned)(rtid + 256 * u);
                asm volatile("" : "+v"(ee));            // keep the index math out of the persistent registers
                const int cl = (int)__umulhi(ee, p.magic_slabW);
                const int tau = (int)ee - cl * p.slabW;
                if (ld_interior) {
                    base0[u] = ((inmask >> u) & 1u) ? 4u * (unsigned)(cl * p.Tin + tau) : 0u;
                } else {
                    const int g = tbase + tau;
                    bool ok = ((inmask >> u) & 1u) && g >= -p.padL && g < hi_lim;
                    int src = g < 0 ? -g : g;
                    src = src >= p.Leff ? refl - src : src;
                    if (p.pad_zero) { src = g; ok = ok && g >= 0; }
                    ok = ok && src < p.Tin;          // zero padding / zero-extension of short inputs (conv.py:89-93)
                    base0[u] = ok ? 4u * (unsigned)(cl * p.Tin + src) : 0u;
                    tile_mask &= ~((ok ? 0u : 1u) << u);
                }
            }
            base_static = ld_interior;
        };
        // loads are unconditional (masked elements read the chunk origin) and issued back to back
        auto load_slab = [&](auto set_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(set_tag)::value;
            const int tbase = ld_tile * BN * p.stride - p.padL;
            if (ld_chunk == 0) setup_tile(tbase);
            const int c0 = ld_chunk * p.CC;
            if (++ld_chunk == p.nchunk) { ld_chunk = 0; ++ld_tile; }
            vmask[S] = tile_mask;
            if (p.ablate & 4) return;
            all_valid[S] = ld_interior;
            // lanes without an element (base0 = 0) read the origin: in bounds, value goes to the dummy slot / is masked
            const unsigned ubase = 4u * (unsigned)(c0 * p.Tin + (ld_interior ? tbase : 0));
            if (p.cin_tail && c0 + p.CC > p.Cin) {      // last chunk runs past the real channels (uniform, rare)
                all_valid[S] = false;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                        unsigned ee = (unsigned)(rtid + 256 * u);
                    asm volatile("" : "+v"(ee));
                    const bool ok = c0 + (int)__umulhi(ee, p.magic_slabW) < p.Cin;
                    const unsigned off = ok ? base0[u] + ubase : 0u;
                    vmask[S] &= ~((ok ? 0u : 1u) << u);
                    v0[S][u] = *(const float*)((const char*)s0b + off);
                    if (DUAL) v1[S][u] = *(const float*)((const char*)s1b + off);
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const unsigned off = base0[u] + ubase;
                v0[S][u] = *(const float*)((const char*)s0b + off);
                if (DUAL) v1[S][u] = *(const float*)((const char*)s1b + off);
            }
        };
        // branch-free per element: lanes without an element write a dummy slot, padding lanes select 0
        // branch-free per element; interior tiles (the common case) skip the padding select
        auto write_slab_t = [&](auto set_tag, char* Xd, auto use_div, auto masked) __attribute__((always_inline)) {
            constexpr int S = decltype(set_tag)::value;
            const int c0 = wr_chunk * p.CC;
            if (++wr_chunk == p.nchunk) wr_chunk = 0;
            const char* t0 = (const char*)(tab0 + c0);
            const char* t1 = (const char*)(tab1 + c0);
            // all table reads first: LDS stores below may alias them as far as the compiler knows, and interleaving
            // would serialise one LDS round trip per element
            float2 a0[PLAIN ? 1 : NU], a1[DUAL ? NU : 1];
            if (!PLAIN) {
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                        a0[u] = *(const float2*)(t0 + cl8[u]);               // a missing element has cl = 0
                    if (DUAL) a1[u] = *(const float2*)(t1 + cl8[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                float v = v0[S][u];
                if (!PLAIN) {
                    if (decltype(use_div)::value) v = v / divv;
                    v = fmaf(v, a0[u].x, a0[u].y);
                    if (DUAL) v = v + fmaf(v1[S][u], a1[u].x, a1[u].y);
                    if (ELU) v = elu_f(v, p.alpha);
                }
                if (decltype(masked)::value) v = ((vmask[S] >> u) & 1u) ? v : 0.f;
                *(float*)(Xd + slot[u]) = v;
            }
        };
        auto write_slab = [&](auto set_tag, char* Xd) __attribute__((always_inline)) {
            constexpr int S = decltype(set_tag)::value;
            if (MODE == 1 && p.div0) write_slab_t(set_tag, Xd, std::true_type(), std::true_type());
            else if (all_valid[S]) write_slab_t(set_tag, Xd, std::false_type(), std::false_type());
            else write_slab_t(set_tag, Xd, std::false_type(), std::true_type());
        };

        using Set0 = std::integral_constant<int, 0>;
        using Set1 = std::integral_constant<int, DEEP ? 1 : 0>;
        int st_tile = t_begin, st_chunk = 0;
        // one pipeline step: (optionally) stream weights of item f+1, turn the registers of item f+1 into its slab, refill
        // that register set with item f+1+NSET, meet the matrix waves at the barrier, reduce a finished tile's statistics
        auto step = [&](int f, auto wr_set) __attribute__((always_inline)) {
            FC_STAMP(1, f, 0);
            if (STAGING_DMA && f + 1 < nitems && !resident) {
                const int nc = st_chunk + 1 == p.nchunk ? 0 : st_chunk + 1;
                dma_weights(wt_tile + (size_t)nc * p.Wbuf, smem + ((f + 1) & 1) * p.Wbuf, p.Wbuf, rtid, p.ablate);
            }
            if (f + 1 < nitems) {
                write_slab(wr_set, (char*)(Xs0 + ((f + 1) & 1) * XSF));
                FC_STAMP(1, f, 1);
                if (f + 1 + NSET < nitems) load_slab(wr_set);
                FC_STAMP(1, f, 2);
            }
            __syncthreads();                          // B(f+1): the matrix waves have finished item f
            FC_STAMP(1, f, 3);
            if (++st_chunk == p.nchunk) { st_chunk = 0; flush_stats(st_tile); ++st_tile; }
            FC_STAMP(1, f, 4);
        };
        load_slab(Set0());
        write_slab(Set0(), (char*)Xs0);
        if (DEEP) {
            if (nitems > 1) load_slab(Set1());        // item 1
            if (nitems > 2) load_slab(Set0());        // item 2
        } else {
            if (nitems > 1) load_slab(Set0());
        }
        __syncthreads();                              // B0: slab 0 + weights 0 visible (the barrier drains the DMA)
        if (DEEP) {
            for (int f = 0; f < nitems; f += 2) {     // item f+1 lives in set 1, item f+2 in set 0
                step(f, Set1());
                if (f + 1 < nitems) step(f + 1, Set0());
            }
        } else {
            for (int f = 0; f < nitems; ++f) step(f, Set0());
        }
        __syncthreads();                              // final (kept symmetric with the matrix role)
        return;
    }

    // =============================================== matrix waves ======================================
    const int wm = wid / WN, wn = wid % WN;
    const int hi = lane >> 5, l31 = lane & 31;
    const int4* kofs = (const int4*)kofs_i;
    f32x16 acc[TM][TN];
    // accumulators start at the bias (the MFMA chain then adds the products): saves one VALU add per output element
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bias = bias_s[(wid / WN) * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][r] = bias;
            }
    };
    zero_acc();

    // epilogue of one tile: bias, store, per-wave GroupNorm partial statistics into red[par][..]
    auto epilogue = [&](int tile, int par) {
        const int n0 = tile * BN;
        int m0_l = m0;                                 // opaque copy keeps the row-pointer math inside the tile loop
        asm volatile("" : "+s"(m0_l));
        const size_t out_off = out_off0;
        float s1 = 0.f, s2 = 0.f;
        const bool full = !p.up_r && p.out_sT == 1 && n0 + BN <= p.Tout && m0_l + BM <= p.M && !(p.ablate & 2) && store_ok;
        if (full) {
            // one 64-bit lane pointer for accumulator row 0; every other row / column tile is a wave-uniform offset
            const size_t sM = (size_t)p.out_sM;
            const float* __restrict__ row0c = p.out + out_off + (size_t)(n0 + wn * (TN * 32) + l31) +
                                             (size_t)(m0_l + wm * (TM * 32) + 4 * hi) * sM;
            float* __restrict__ row0 = const_cast<float*>(row0c);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* __restrict__ rowp = row0 + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * sM;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float v = acc[i][j][r];
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                        rowp[j * 32] = v;
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const int m = m0_l + ml;
                    if (m >= p.M) continue;
                    int co = m, phs = 0;
                    if (p.up_r > 1) { co = (int)__umulhi((unsigned)m, p.magic_r); phs = m - co * p.up_r; }   // up_r == 1: co = m (k = 2, stride 1)
                    float* __restrict__ rowp = p.out + out_off + (size_t)co * p.out_sM;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int n = n0 + wn * (TN * 32) + j * 32 + l31;
                        if (n >= p.Tout) continue;
                        const float v = acc[i][j][r];
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                        if ((p.ablate & 2) || !store_ok) continue;
                        if (p.up_r) {
                            const int t = n * p.up_r + phs - p.trimL;
                            if (t >= 0 && t < p.Tfinal) rowp[t] = v;
                        } else {
                            rowp[(size_t)n * p.out_sT] = v;
                        }
                    }
                }
            }
        }
        // per-lane partials go to LDS as they are; the STAGING waves (which have slack) reduce them after the next
        // barrier, so the matrix waves never pay for the cross-lane reduction
        if (p.partials) red[par * 256 + rtid] = make_float2(s1, s2);
    };
    if (!STAGING_DMA) {
        dma_weights(wt_tile, smem, p.Wbuf, rtid, p.ablate);
        if (resident && p.nchunk == 2) dma_weights(wt_tile + p.Wbuf, smem + p.Wbuf, p.Wbuf, rtid, p.ablate);
    }
    __syncthreads();                                  // B0 (drains the weight DMA)

    const int a_off = hi * BM + wm * (TM * 32) + l31;
    const int b_off = hi * p.rowStride + wn * (TN * 32) + l31;
    const int nks2 = ((p.Kc >> 1) + 3) >> 2 << 1;     // groups of 2 k-steps, rounded up to pairs of groups
    int tile = t_begin, chunk = 0;
    for (int f = 0; f < nitems; ++f) {
        FC_STAMP(0, f, 0);
        FC_STAMP(0, f, 7);
        // the next weight chunk streams in as 1 KiB DMA pieces issued BETWEEN MFMA groups (one per loop trip): a piece
        // costs the issuing wave ~100 cycles, which is free while its previous MFMAs are still executing but not when
        // all pieces are issued back to back ahead of the loop
        const bool stream_w = !STAGING_DMA && f + 1 < nitems && !resident && !(p.ablate & 16);
        const int nc = chunk + 1 == p.nchunk ? 0 : chunk + 1;
        const float* wsrc = wt_tile + (size_t)nc * p.Wbuf + rtid * 4;
        float* wdst = smem + ((f + 1) & 1) * p.Wbuf + __builtin_amdgcn_readfirstlane(wid) * 256;
        int wleft = stream_w ? p.Wbuf : 0;            // floats still to request (1024 per piece over the 4 waves)
        const float* Ws = smem + (resident ? chunk : (f & 1)) * p.Wbuf + a_off;
        const float* Xb = Xs0 + (f & 1) * XSF + b_off;
        // k-steps (2 k values each): kidx = 2*ks + hi = kk*CC + 2*c2 + hi, in groups of two.  The loop is software
        // pipelined by hand: the LDS fragment reads (and the B-offset table entry) of group g+1 are issued BEFORE the 8
        // MFMAs of group g and the scheduler is fenced so that it keeps them there -- left alone hipcc sinks every read
        // to just above its first use and each MFMA quad then eats a full LDS round trip (visible whenever fewer than
        // ~3 matrix waves share a SIMD).  Reads past the chunk's last k-step stay inside LDS and are never used.
        auto load_group = [&](int g, const int2 k2, float (&a)[2][TM], float (&bb)[2][TN]) {
            const int kos[2] = {k2.x, k2.y};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[u][i] = Ws[(g * 2 + u) * 2 * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[u][j] = Xb[kos[u] + j * 32];
            }
        };
        auto mfma_group = [&](const float (&a)[2][TM], const float (&bb)[2][TN]) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], bb[u][j], acc[i][j], 0, 0, 0);
        };
        if (!(p.ablate & 1)) {
            // the packed weight image is zero-padded to a multiple of 4 k-steps (conv_wbuf_floats) and the offset
            // table to two groups more, so there is no tail: padded k-steps multiply zeros into the accumulators
            const int2* kofs2 = (const int2*)kofs;
            float fa0[2][TM], fb0[2][TN], fa1[2][TM], fb1[2][TN];
            load_group(0, kofs2[0], fa0, fb0);
            int2 ko = kofs2[1];
            for (int g = 0; g < nks2; g += 2) {
                load_group(g + 1, ko, fa1, fb1);
                ko = kofs2[g + 2];
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                if (wleft > 0) {
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)wsrc, (lds_ptr_t)wdst, 16, 0, 0);
                    wsrc += 1024; wdst += 1024; wleft -= 1024;
                }
                load_group(g + 2, ko, fa0, fb0);
                ko = kofs2[g + 3];
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (; wleft > 0; wleft -= 1024, wsrc += 1024, wdst += 1024)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)wsrc, (lds_ptr_t)wdst, 16, 0, 0);
        FC_STAMP(0, f, 1);
        const bool tile_done = chunk == p.nchunk - 1;
        if (tile_done) {
            if (!(p.ablate & 128)) epilogue(tile, tile & 1);
            zero_acc();
        }
        FC_STAMP(0, f, 2);
        __syncthreads();                              // B(f+1): slab f+1 + weights f+1 visible, buffers f free
        FC_STAMP(0, f, 3);
        if (tile_done) { ++tile; chunk = 0; } else { ++chunk; }
    }
    __syncthreads();                                  // final: publishes the last tile's per-lane partials
}

// entries of the k-step offset table: the k-steps of a chunk rounded up to whole groups of 4, plus two zero groups the
// pipelined main loop may read ahead
template <int BM, int BN, int WM, int WN, int MODE, int NU, bool ROW>
static hipError_t launch_conv_k(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    // the opt-in to > 64 KiB of dynamic LDS is per (kernel, device): one bit per device, set idempotently (two threads
    // racing here both set the attribute, which is harmless)
    static std::atomic<unsigned long long> attr_done{0ull};
    auto kfn = conv_mfma_kernel<BM, BN, WM, WN, MODE, NU, ROW>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        hipError_t ea = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return ea;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, a);
    return hipGetLastError();
}

// slots of the instantiation that stages `total` slab elements (0: none)
static int conv_nu_for(int total, int mode) {
    const int need = (total + 255) / 256;
    if (need <= 5) return 5;
    if (need <= 9) return 9;
    if (mode >= 3) return 0;
    if (mode == 0 && need <= 11) return 11;
    if (mode == 0 && need <= 16) return 16;
    return need <= NU_BIG ? NU_BIG : 0;
}

template <int BM, int BN, int WM, int WN, int MODE>
static hipError_t launch_conv_m(const ConvArgs& a, int total, dim3 grid, size_t lds, hipStream_t st) {
    if (a.row) {   // row staging: NU = rounds per item (2 / 4 / 8 specialised, 1 = run-time count)
        const int nr = a.CC / (4 * (64 / (BN / 4)));
        if (nr == 2) return launch_conv_k<BM, BN, WM, WN, MODE, 2, true>(a, grid, lds, st);
        if (nr == 4) return launch_conv_k<BM, BN, WM, WN, MODE, 4, true>(a, grid, lds, st);
        if constexpr (MODE < 3) {
            if (nr == 8) return launch_conv_k<BM, BN, WM, WN, MODE, 8, true>(a, grid, lds, st);
        }
        return launch_conv_k<BM, BN, WM, WN, MODE, 1, true>(a, grid, lds, st);
    }
    const int nu = conv_nu_for(total, MODE);
    if (nu == 5) return launch_conv_k<BM, BN, WM, WN, MODE, 5, false>(a, grid, lds, st);
    if (nu == 9) return launch_conv_k<BM, BN, WM, WN, MODE, 9, false>(a, grid, lds, st);
    if constexpr (MODE < 3) {
        if constexpr (MODE == 0) {
            if (nu == 11) return launch_conv_k<BM, BN, WM, WN, MODE, 11, false>(a, grid, lds, st);
            if (nu == 16) return launch_conv_k<BM, BN, WM, WN, MODE, 16, false>(a, grid, lds, st);
        }
        if (nu == NU_BIG) return launch_conv_k<BM, BN, WM, WN, MODE, NU_BIG, false>(a, grid, lds, st);
    }
    return hipErrorInvalidValue;
}

// one explicit instantiation per tile shape, each in its own translation unit (conv_tile_*.hip) so that they compile in parallel
template <int BM, int BN, int WM, int WN>
hipError_t launch_conv_tile(const ConvLaunch& c, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    const int total = a.CC * a.rowStride;
    if (c.s1.ptr) return c.elu ? launch_conv_m<BM, BN, WM, WN, 4>(a, total, grid, lds, st)
                               : launch_conv_m<BM, BN, WM, WN, 3>(a, total, grid, lds, st);
    if (c.s0.aff || c.s0.div || c.elu) return c.elu ? launch_conv_m<BM, BN, WM, WN, 2>(a, total, grid, lds, st)
                                                    : launch_conv_m<BM, BN, WM, WN, 1>(a, total, grid, lds, st);
    return launch_conv_m<BM, BN, WM, WN, 0>(a, total, grid, lds, st);
}

// which template instantiation launch_conv() will pick (profiling labels)
}  // namespace fc
