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
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));     // 4-byte aligned vector stores (global memory takes them)
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));

static inline __host__ __device__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Ablation masks (ConvArgs::ablate and friends) exist in tuning builds only (FC_BUILD_DEFINES=FC_AB_KNOBS / FC_TIMELINE): in the shipped
// library FC_ABL() is the constant 0, so no profiling branch sits around a load, a DMA piece or an MFMA block of the hot loops (a run-time
// condition around a software-pipelined load is exactly what hipcc turns into exposed latency).
#ifndef FC_ELEM_NSET
#define FC_ELEM_NSET 1      // ... of the quad element staging of the prologue modes (plain layers always keep two); 2 measured: slower (below)
#endif
#ifndef FC_ROW_NSET
#define FC_ROW_NSET 1       // register sets of the quad row staging (A / B builds: FC_BUILD_DEFINES="FC_ROW_NSET=2 FC_ELEM_NSET=2")
#endif
#ifdef FC_AB_KNOBS
#define FC_ABL(mask_, bits_) ((mask_) & (bits_))
#else
#define FC_ABL(mask_, bits_) (0)
#endif

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
//          QUAD-K form (round 5, QK = true: every chunking of a multiple of 4 channels, i.e. all but a net's first conv):
//          Ws[2][quad][hi][BM][4]  a lane's weights of FOUR consecutive k-steps are one 16-byte piece (kernels.hip conv_pack_index)
//          Xs[2][CC/4][S][PL][4]   slab planes of 4 channels, a tap = a column offset; half-quad h = tap * CC/4 + channel quad, quad q =
//                          half-quads 2q (lanes 0-31) and 2q + 1 (lanes 32-63); one ds_read_b128 per operand tile per four k-steps
//          MODE 5: the input arrives materialised as xq[b][C/4][padL + T + padR][4] (combine_xq_kernel) and slab + weights are staged
//                          by LDS-DMA pieces with scalar bases: no vector instruction in the staging waves
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
    int xq_Tp;              // > 0: src0 is a channel-quad-interleaved, padded materialisation xq[b][Cin/4][xq_Tp][4] (combine_xq_kernel;
                            // column t of the tensor sits at xq column padL + t, the padding values are already in place): MODE 5 DMA staging
    int quad;               // quad-k operand layout (CC % 4 == 0): A [quad][hi][BM][4], B [g4][column][4 channels]
    int nq2;                // quads per chunk, rounded up to an even count
    int nq_odd;             // 1: the last quad of the pair count is padding only (k * CC / 4 half-quads = an odd number of quads): its MFMAs are skipped
    int cin_tail;           // Cin % CC != 0: the last chunk runs past the real channels
    // two-level batch addressing (2-D nets in frequency-major layout [B][F][C][T]: a frequency row of an utterance is one
    // "virtual utterance" of the 1-D conv): blockIdx.z = breal*Fo + fo.  1-D layers: Fo = 1, in_sB0 = Cin*Tin, affC = Cin.
    int Fo;                 // virtual utterances (output frequency rows) per real utterance
    int affC;               // channels of the affine tables (table index = channel % affC)
    long long in_sB0, in_sB1;     // floats between real utterances / consecutive fo of the inputs
    long long out_sF;             // floats between consecutive fo of the output (out_sB: between real utterances)
    long long part_sB0;           // partial (sum, sumsq) pairs between real utterances; fo-th row at fo * nblk
    int store_lo, store_hi;       // only virtual utterances fo in [store_lo, store_hi) store their outputs (all contribute statistics)
    int ablate;             // profiling aid (FC_ABLATE env, FC_AB_KNOBS builds only): 1 no MFMA, 2 no stores, 4 no slab loads, 16 no weight DMA,
                            // 128 no epilogue, 256 no prologue arithmetic (quad staging paths), 512 no slab write at all (quad staging paths)
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

// Buffer loads for the register-staged quad paths (round 5): `buffer_load_dword v, v_off, s[rsrc], s_off offen` forms its address from a scalar
// descriptor, a scalar offset and ONE 32-bit lane offset -- no vector instruction.  The global_load form hipcc picks for `base + lane offset`
// computes a 64-bit lane address per load (v_lshl_add_u64: 114 of them in the staging loop of the two-source strided layers), and a staging
// wave's vector instructions are what it cannot get issued next to the matrix waves' MFMA stream.
typedef unsigned int u32x4_t __attribute__((__vector_size__(16)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stage_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);   // raw buffer, range check effectively off
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
typedef float f32x2_ld __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_ld buf_ld2(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x2_ld, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, 0));
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
    if (FC_ABL(ablate, 16)) return;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int off = 0; off < nfloats; off += 1024) {
        const float* g = gsrc + off + tid * 4;
        float* l = lds_dst + off + wid * 256;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
    }
}


// One LDS-DMA piece (64 lanes x 16 bytes -> 1 KiB of LDS at the wave-uniform byte address `lds_byte`) whose 64 source addresses are a
// wave-uniform 64-bit base (SGPR pair) + a per-lane 32-bit byte offset: NO vector ALU instruction is needed to form the addresses.  That is
// the point (round 5): fp32 MFMAs occupy the SIMD's vector issue, and a co-resident wave's VALU instruction waits for a gap in the matrix
// wave's MFMA stream -- the builtin form computes a 64-bit lane address per piece (v_lshl_add_u64) and the staging wave of a DMA-only item
// still took a whole item to get its ~10 pieces issued.  M0 is compiler-reserved: saved and restored inside the statement.  hipcc does
// not count this load: the caller waits with dma_wait_all() before the barrier that publishes the data.
__device__ __forceinline__ void dma_piece_sbase(const void* sbase, unsigned voff, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
template <typename T>
__device__ __forceinline__ const T* sgpr_ptr(const T* q) {        // a wave-uniform pointer, pinned to a scalar register pair
    const unsigned long long v = (unsigned long long)(size_t)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const T*)(size_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_byte_addr(const float* p) { return (unsigned)(size_t)(lds_ptr_t)p; }

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
// QK: quad-k operand layout (round 5).  Both operands are stored so that a lane's values of FOUR consecutive k-steps are one 16-byte piece:
//   Ws[quad][hi][BM][4]          a chunk's k indices as HALF-QUADS h = (tap kk, channel quad g4) in tap-major order; quad q = half-quads 2 q
//                                (MFMA lanes 0-31) and 2 q + 1 (lanes 32-63); element s = weight of channel 4 g4 + s at tap kk
//   Xs[g4][slab column][4]       element s = channel 4 g4 + s of that column; a tap is a column offset, a stride phase a column block; the
//                                lane halves read their half-quad's (plane, column) offset from table[2 q + hi]
// so the matrix waves issue ONE ds_read_b128 per operand tile per 4 k-steps (16 MFMAs of a 2 x 2 register tile) instead of one ds_read_b32 per
// tile per k-step: tests/micro/conv_loop_feed_b128.hip measures 0.95 - 0.98 of the fp32 MFMA peak for this feed against 0.81 (one matrix wave
// per SIMD) / 0.89 (two) for the round-4 form.  k-step s of a quad multiplies channel s of its two half-quads: the accumulation order inside
// a chunk is (quad, s), fixed per layer.  NU then counts QUAD slots (4 channels x one column) per staging
// thread (element staging) or rounds of 256 (4 channels x 4 columns) units (row staging).
template <int BM, int BN, int WM, int WN, int MODE, int NU, bool ROW, bool QK = false>
__global__ __launch_bounds__(512, 4) void conv_mfma_kernel(const ConvArgs p) {
    static_assert(WM * WN == 4, "4 matrix waves per workgroup");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    // MODE: 0 plain | 1 affine | 2 affine+ELU | 3 dual | 4 dual+ELU | 5 plain, slab staged by DMA from a quad-interleaved padded input
    constexpr bool PLAIN = MODE == 0 || MODE == 5;
    constexpr bool DMA5 = MODE == 5;
    static_assert(!DMA5 || QK, "DMA slab staging needs the quad layout");
    constexpr bool DUAL = MODE == 3 || MODE == 4;
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
        if constexpr (DMA5) {
            // ---------------------------------------------------------------------------------------------
            // MODE 5 (round 5): the slab arrives by DMA, like the weights.  The timelines of round 5 showed what bounds
            // this kernel: fp32 MFMAs occupy the SIMD's vector issue, and every VALU instruction of the co-resident staging wave waits for a
            // gap in the matrix wave's MFMA stream -- the staging waves of the register-staged forms took a whole item (~9 000 cycles) to get
            // ~100 instructions issued, and the matrix waves then waited for them at the barrier.  Here the input was materialised (it is, for
            // every layer with >= 3 M tiles) as xq[b][channel quad][padded column][4 channels], so a B-operand plane row is contiguous in
            // memory: a staging wave issues NU 16-byte-per-lane DMA pieces per item (1 KiB each, no VGPR data, no prologue, no LDS store, no
            // edge cases: the padding is part of the materialisation) and goes back to the barrier.
            // piece d = (j * 4 + wave) * 64 + lane of the slab image [g4][rowStride columns]; its source column: stride phases are column
            // blocks of PL (col = ph * PL + q <-> input column q * stride + ph)
            // ---------------------------------------------------------------------------------------------
            const int pieces = (p.CC >> 2) * p.rowStride;
            unsigned off[NU];
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const int d = ((j * 4 + wid) << 6) + lane;
                off[j] = 0u;
                if (d < pieces) {
                    const int g4 = d / p.rowStride, col = d - g4 * p.rowStride;
                    const int ph = col / p.PL, q = col - ph * p.PL;
                    off[j] = 16u * (unsigned)(g4 * p.xq_Tp + q * p.stride + ph);
                }
            }
            // the piece form below needs its 64-bit bases in SGPRs: pinned (sgpr_ptr) once per item in dma_w / dma_slab -- when the epilogue's
            // address arithmetic grows, hipcc otherwise moves the whole chain that starts at the utterance index to vector registers and
            // hands the asm a VGPR pair ("invalid operand"); on an SGPR value the two v_readfirstlane fold away
            const char* xq_b = (const char*)p.src0 + (size_t)breal * (size_t)(p.Cin >> 2) * (size_t)p.xq_Tp * 16;
            const size_t chunk_bytes = (size_t)(p.CC >> 2) * (size_t)p.xq_Tp * 16;
            const int wid_s = __builtin_amdgcn_readfirstlane(wid);
            const unsigned w_voff = 16u * (unsigned)rtid;         // weights: piece i of a chunk = 4 KiB, this wave's KiB of it at wave * 1 KiB
            const unsigned smem_b = lds_byte_addr(smem), xs0_b = lds_byte_addr(Xs0);
            const int wpieces = p.Wbuf >> 10;
            int ld_tile = t_begin, ld_chunk = 0;
            // the staging waves issue nothing but scalar and memory instructions from here on; they must still win the issue arbitration
            // against the matrix waves whenever they have something to issue (a DMA requested late is a barrier wait for four matrix waves)
            __builtin_amdgcn_s_setprio(3);
            auto dma_w = [&](const float* gsrc, int buf) __attribute__((always_inline)) {
                if FC_ABL(p.ablate, 16) return;
                const char* g = sgpr_ptr((const char*)gsrc);
                unsigned l = smem_b + (unsigned)buf * (unsigned)p.Wbuf * 4u + (unsigned)wid_s * 1024u;
                for (int i = 0; i < wpieces; ++i, g += 4096, l += 4096u) dma_piece_sbase(g, w_voff, l);
            };
            auto dma_slab = [&](int buf) __attribute__((always_inline)) {
                const char* base = sgpr_ptr(xq_b + (size_t)ld_chunk * chunk_bytes + (size_t)ld_tile * (size_t)(BN * p.stride) * 16);
                if (++ld_chunk == p.nchunk) { ld_chunk = 0; ++ld_tile; }
                if FC_ABL(p.ablate, 4) return;
                const unsigned l = xs0_b + (unsigned)buf * (unsigned)XSF * 4u + (unsigned)wid_s * 1024u;
#pragma unroll
                for (int j = 0; j < NU; ++j) dma_piece_sbase(base, off[j], l + (unsigned)j * 4096u);
            };
            dma_slab(0);                                  // item 0 (its weights were requested above, by the builtin form hipcc counts)
            dma_wait_all();
            __syncthreads();                              // B0
            int st_tile = t_begin, st_chunk = 0;
            for (int f = 0; f < nitems; ++f) {
                FC_STAMP(1, f, 0);
                if (f + 1 < nitems) {
                    if (!resident) {
                        const int nc = st_chunk + 1 == p.nchunk ? 0 : st_chunk + 1;
                        dma_w(wt_tile + (size_t)nc * p.Wbuf, (f + 1) & 1);
                    }
                    dma_slab((f + 1) & 1);
                }
                FC_STAMP(1, f, 1);
                dma_wait_all();                           // hipcc does not count the asm pieces: the barrier below publishes them
                FC_STAMP(1, f, 2);
                __syncthreads();                          // B(f+1)
                FC_STAMP(1, f, 3);
                if (++st_chunk == p.nchunk) { st_chunk = 0; flush_stats(st_tile); ++st_tile; }
                FC_STAMP(1, f, 4);
            }
            __syncthreads();                              // final (kept symmetric with the matrix role)
            return;
        }
        if constexpr (ROW && QK) {
            // ---------------------------------------------------------------------------------------------
            // Row staging, quad-k layout (stride-1 layers).  A unit = 4 consecutive channels x CW consecutive columns (CW = 4): FOUR 16-byte
            // global loads (one per channel) and FOUR 16-byte LDS stores (one per column: the 4 channels of that column are one B-operand
            // piece); the 4 x 4 transposition is free, it only names registers.  BN/CW threads cover the main columns of a channel quad, a
            // round = 256 units = 256/(BN/CW) channel quads; NU rounds per item (compile time).  Chunks with fewer than 256 such units take
            // NARROWER units (NU = 12: CW = 2, NU = 11: CW = 1; one round) so that all four staging waves share the work -- a staging wave's
            // instructions are issued in the gaps of the co-resident matrix wave's MFMA stream, the item lasts as long as the busiest one.
            // Tail columns ((k - 1) * dilation per row): one (channel quad, column) unit per thread.
            // ---------------------------------------------------------------------------------------------
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
            // NU 14 / 13 (round 6): ONE full round of 256 (4 channels x 4 columns) units re-cut as 4 rounds of 1-column units / 2 rounds of 2-column
            // units.  Why: a `ds_write_b128` is serviced in groups of 8 consecutive lanes over 32 banks (MI355X_MICROARCH.md), and with 4 columns
            // per lane the lanes of a group are 64 bytes apart -- two distinct 16-byte slots for eight lanes, a 4-way bank conflict on every slab
            // store (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.35 - 0.38 on the k = 1 row-staged classes, 0.00 on the DMA-staged ones,
            // profiles/r06_pmc_step.txt).  One column per lane makes the stores conflict-free at the price of dword instead of 16-byte loads.
            constexpr int CW = (NU == 12 || NU == 13) ? 2 : ((NU == 11 || NU == 14) ? 1 : 4), NR = NU == 14 ? 4 : (NU == 13 ? 2 : (NU >= 11 ? 1 : NU));
            constexpr int LPR = BN / CW, GPR = 256 / LPR;
            const int nq4 = p.CC >> 2;
            const bool has_main = rtid / LPR < nq4;        // wave-uniform (LPR >= 32); false only when the chunk has < 256 units (NR == 1)
            const int g0 = has_main ? rtid / LPR : 0, c4 = rtid % LPR;
            const unsigned slot0 = 16u * (unsigned)(g0 * p.rowStride + CW * c4);
            const unsigned lds_round = 16u * (unsigned)(GPR * p.rowStride);
            const size_t src_round = (size_t)(4 * GPR) * p.Tin;
            const int km1 = p.slabW - BN;
            const bool has_tail = rtid < nq4 * km1;
            const int t_g = has_tail ? rtid / km1 : 0, t_j = has_tail ? rtid - t_g * km1 : 0;
            const unsigned t_slot = 16u * (unsigned)(t_g * p.rowStride + BN + t_j);
            unsigned src_off = 0, eoff[CW], emask = 0, t_off = 0;
            bool ld_edge = false, t_ok = true;
            // Register sets in flight (FC_ROW_NSET, default 1).  Round 6 measured what the fused-prologue classes wait for
            // (profiles/r06_prologue_ablation.txt, profiles/r06_two_register_sets.txt): with the prologue ARITHMETIC removed they do not move
            // (11.51 vs 11.47 ms of conv time per step), with the slab WRITE removed altogether they gain 5 - 17 % (0.51 ms per step in total:
            // the upper bound of any staging rework), and a second register set (loads of item f + 3 in flight while item f + 2's values wait)
            // returns nothing on the row-staged classes and loses 4 - 12 % on the element-staged ones (8 spilled registers): 15.04 vs 14.96 ms.
            constexpr int NSET = FC_ROW_NSET;
            bool r_edge[NSET] = {}, r_tok[NSET] = {}; unsigned r_emask[NSET] = {};
            f32x4 v0[NSET][NR][4], v1[NSET][DUAL ? NR : 1][4];
            float tv0[NSET][4] = {}, tv1[NSET][4] = {};
            int ld_tile = t_begin, ld_chunk = 0, wr_chunk = 0;
            const int hi_lim = p.Tin + p.padR, refl = 2 * (p.Leff - 1);
            auto resolve = [&](int g, bool& ok) __attribute__((always_inline)) {
                ok = g >= -p.padL && g < hi_lim;
                int src = g < 0 ? -g : g;
                src = src >= p.Leff ? refl - src : src;
                if (p.pad_zero) { src = g; ok = ok && g >= 0; }
                ok = ok && src < p.Tin;
                return ok ? src : 0;
            };
            auto setup_tile = [&](int tbase) __attribute__((always_inline)) {
                ld_edge = !(tbase >= 0 && tbase + p.slabW <= p.Tin);
                if (!ld_edge) {
                    src_off = 4u * (unsigned)(4 * g0 * p.Tin + tbase + CW * c4);
                    t_off = has_tail ? 4u * (unsigned)(4 * t_g * p.Tin + tbase + BN + t_j) : src_off;
                    t_ok = true;
                } else {
                    emask = 0;
#pragma unroll
                    for (int j = 0; j < CW; ++j) {
                        bool ok;
                        const int src = resolve(tbase + CW * c4 + j, ok);
                        eoff[j] = 4u * (unsigned)(4 * g0 * p.Tin + src);
                        emask |= (ok ? 1u : 0u) << j;
                    }
                    const int src = resolve(has_tail ? tbase + BN + t_j : tbase + CW * c4, t_ok);
                    t_off = 4u * (unsigned)(4 * (has_tail ? t_g : g0) * p.Tin + src);
                }
            };
            auto load_slab = [&](auto set_tag) __attribute__((always_inline)) {
                constexpr int S = decltype(set_tag)::value;
                const int tbase = ld_tile * BN - p.padL;
                if (ld_chunk == 0) setup_tile(tbase);
                const size_t cbase = (size_t)(ld_chunk * p.CC) * p.Tin;
                if (++ld_chunk == p.nchunk) { ld_chunk = 0; ++ld_tile; }
                r_edge[S] = ld_edge; r_emask[S] = emask; r_tok[S] = t_ok;
                if FC_ABL(p.ablate, 4) return;
                // the chunk's rows through buffer descriptors: scalar base + scalar (round, channel) offset + the lane's 32-bit offset
                const __amdgpu_buffer_rsrc_t q0 = stage_rsrc(s0b + cbase), q1 = stage_rsrc(s1b + cbase);
                const int ch_b = 4 * p.Tin, rd_b = 4 * (int)src_round;       // bytes between channels / rounds (< 2^31: launch_conv checks CC * Tin)
                if (!ld_edge) {
#pragma unroll
                    for (int r = 0; r < NR; ++r)
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            const int so = r * rd_b + s4 * ch_b;
                            if constexpr (CW == 4) {
                                v0[S][r][s4] = buf_ld4(q0, src_off, so);
                                if (DUAL) v1[S][r][s4] = buf_ld4(q1, src_off, so);
                            } else if constexpr (CW == 2) {
                                const f32x2_ld t0 = buf_ld2(q0, src_off, so);
                                v0[S][r][s4][0] = t0[0]; v0[S][r][s4][1] = t0[1];
                                if (DUAL) { const f32x2_ld t1 = buf_ld2(q1, src_off, so); v1[S][r][s4][0] = t1[0]; v1[S][r][s4][1] = t1[1]; }
                            } else {
                                v0[S][r][s4][0] = buf_ld1(q0, src_off, so);
                                if (DUAL) v1[S][r][s4][0] = buf_ld1(q1, src_off, so);
                            }
                        }
                } else {
#pragma unroll
                    for (int r = 0; r < NR; ++r)
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                            for (int j = 0; j < CW; ++j) {
                                v0[S][r][s4][j] = buf_ld1(q0, eoff[j], r * rd_b + s4 * ch_b);
                                if (DUAL) v1[S][r][s4][j] = buf_ld1(q1, eoff[j], r * rd_b + s4 * ch_b);
                            }
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    tv0[S][s4] = buf_ld1(q0, t_off, s4 * ch_b);
                    if (DUAL) tv1[S][s4] = buf_ld1(q1, t_off, s4 * ch_b);
                }
            };
            auto prologue = [&](float v, float w, float2 a, float2 a1) __attribute__((always_inline)) {
                if (PLAIN || FC_ABL(p.ablate, 256)) return v;
                v = fmaf(v, a.x, a.y);
                if (DUAL) v = v + fmaf(w, a1.x, a1.y);
                if (ELU) v = elu_f(v, p.alpha);
                return v;
            };
            // the 4 table entries (scale, shift) of a channel quad: 32 contiguous bytes
            auto tab4 = [&](const float2* t, int c, float2 (&a)[4]) __attribute__((always_inline)) {
                const float4 lo = *(const float4*)(t + c), hi4 = *(const float4*)(t + c + 2);
                a[0] = make_float2(lo.x, lo.y); a[1] = make_float2(lo.z, lo.w);
                a[2] = make_float2(hi4.x, hi4.y); a[3] = make_float2(hi4.z, hi4.w);
            };
            auto write_slab = [&](auto set_tag, char* Xd) __attribute__((always_inline)) {
                constexpr int S = decltype(set_tag)::value;
                const int c0 = wr_chunk * p.CC;
                if (++wr_chunk == p.nchunk) wr_chunk = 0;
                if (FC_ABL(p.ablate, 512)) return;
                float2 a[PLAIN ? 1 : NR][4], a1[DUAL ? NR : 1][4], ta[4], ta1[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { ta[s4] = make_float2(1.f, 0.f); ta1[s4] = make_float2(1.f, 0.f); }
                if (!PLAIN) {                             // all table reads ahead of the stores (possible aliasing)
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        tab4(tab0, c0 + 4 * (g0 + r * GPR), a[r]);
                        if (DUAL) tab4(tab1, c0 + 4 * (g0 + r * GPR), a1[r]);
                    }
                    tab4(tab0, c0 + 4 * t_g, ta);
                    if (DUAL) tab4(tab1, c0 + 4 * t_g, ta1);
                }
                if (has_main) {
#pragma unroll
                    for (int r = 0; r < NR; ++r)
#pragma unroll
                        for (int j = 0; j < CW; ++j) {
                            f32x4 o;
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                o[s4] = prologue(v0[S][r][s4][j], DUAL ? v1[S][r][s4][j] : 0.f, PLAIN ? ta[s4] : a[r][s4], DUAL ? a1[r][s4] : ta1[s4]);
                                if (r_edge[S]) o[s4] = ((r_emask[S] >> j) & 1u) ? o[s4] : 0.f;
                            }
                            *(f32x4*)(Xd + slot0 + r * lds_round + 16 * j) = o;
                        }
                }
                if (has_tail) {
                    f32x4 o;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        o[s4] = prologue(tv0[S][s4], tv1[S][s4], ta[s4], ta1[s4]);
                        o[s4] = r_tok[S] ? o[s4] : 0.f;
                    }
                    *(f32x4*)(Xd + t_slot) = o;
                }
            };
            using Set0 = std::integral_constant<int, 0>;
            using Set1 = std::integral_constant<int, NSET == 2 ? 1 : 0>;
            int st_tile = t_begin, st_chunk = 0;
            // one pipeline step: item f + 1 (held in register set wr_set) becomes its slab, that set is refilled with item f + 1 + NSET
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
                __syncthreads();                          // B(f+1)
                FC_STAMP(1, f, 3);
                if (++st_chunk == p.nchunk) { st_chunk = 0; flush_stats(st_tile); ++st_tile; }
                FC_STAMP(1, f, 4);
            };
            load_slab(Set0());
            write_slab(Set0(), (char*)Xs0);
            if (NSET == 2) {
                if (nitems > 1) load_slab(Set1());        // item 1
                if (nitems > 2) load_slab(Set0());        // item 2
            } else {
                if (nitems > 1) load_slab(Set0());
            }
            __syncthreads();                              // B0
            if (NSET == 2) {
                for (int f = 0; f < nitems; f += 2) {     // item f + 1 lives in set 1, item f + 2 in set 0
                    step(f, Set1());
                    if (f + 1 < nitems) step(f + 1, Set0());
                }
            } else {
                for (int f = 0; f < nitems; ++f) step(f, Set0());
            }
            __syncthreads();                              // final (kept symmetric with the matrix role)
            return;
        }
        if constexpr (ROW && !QK) {
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
                if FC_ABL(p.ablate, 4) return;
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
                if (PLAIN || FC_ABL(p.ablate, 256)) return v;
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
        if constexpr (QK) {
            // ---------------------------------------------------------------------------------------------
            // Element staging, quad-k layout (strided layers, PLAIN stride-1 layers): a slot = (channel quad, slab column): FOUR dword loads
            // (the 4 channels: same lane offset from 4 wave-uniform row bases) and ONE 16-byte LDS store into the column's B-operand piece
            // (stride-phase-split like the round-4 slab: [c8][hi][tau % stride][tau / stride][4]).  NU = slots per thread (compile time).
            // ---------------------------------------------------------------------------------------------
            const int totalq = (p.CC >> 2) * p.slabW;
            const float divv = (MODE == 1 && p.div0) ? p.div0[breal] : 1.f;
            unsigned base0[NU], slot[NU], cl32[PLAIN ? 1 : NU];
            constexpr bool DEEPQ = DEEP || FC_ELEM_NSET == 2;     // round 6: two register sets for the prologue modes as well (see the row staging above)
            constexpr int NSET = DEEPQ ? 2 : 1;
            float v0[NSET][NU][4], v1[NSET][DUAL ? NU : 1][4];
            unsigned inmask = 0, vmask[NSET] = {};
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int e = rtid + 256 * u;
                base0[u] = 0u; slot[u] = (unsigned)(XSF - 4) * 4u;
                if (!PLAIN) cl32[u] = 0u;
                if (e < totalq) {
                    const int g4 = (int)__umulhi((unsigned)e, p.magic_slabW);
                    const int tau = e - g4 * p.slabW;
                    int ph, q;
                    switch (p.stride) {
                        case 1: ph = 0; q = tau; break;
                        case 2: q = tau >> 1; ph = tau & 1; break;
                        case 4: q = tau >> 2; ph = tau & 3; break;
                        case 8: q = tau >> 3; ph = tau & 7; break;
                        default: q = tau / p.stride; ph = tau - q * p.stride; break;
                    }
                    base0[u] = 4u * (unsigned)(4 * g4 * p.Tin + tau);
                    slot[u] = 16u * (unsigned)(g4 * p.rowStride + ph * p.PL + q);
                    if (!PLAIN) cl32[u] = (unsigned)g4 * 32u;
                    inmask |= 1u << u;
                }
            }
            bool base_static = true, ld_interior = true;
            unsigned tile_mask = 0;
            bool all_valid[NSET] = {};
            int ld_tile = t_begin, ld_chunk = 0, wr_chunk = 0;
            auto setup_tile = [&](int tbase) __attribute__((always_inline)) {
                ld_interior = tbase >= 0 && tbase + p.slabW <= p.Tin;
                tile_mask = inmask;
                if (ld_interior && base_static) return;
                const int hi_lim = p.Tin + p.padR, refl = 2 * (p.Leff - 1);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    unsigned ee = (unsigned)(rtid + 256 * u);
                    asm volatile("" : "+v"(ee));
                    const int g4 = (int)__umulhi(ee, p.magic_slabW);
                    const int tau = (int)ee - g4 * p.slabW;
                    if (ld_interior) {
                        base0[u] = ((inmask >> u) & 1u) ? 4u * (unsigned)(4 * g4 * p.Tin + tau) : 0u;
                    } else {
                        const int g = tbase + tau;
                        bool ok = ((inmask >> u) & 1u) && g >= -p.padL && g < hi_lim;
                        int src = g < 0 ? -g : g;
                        src = src >= p.Leff ? refl - src : src;
                        if (p.pad_zero) { src = g; ok = ok && g >= 0; }
                        ok = ok && src < p.Tin;
                        base0[u] = ok ? 4u * (unsigned)(4 * g4 * p.Tin + src) : 0u;
                        tile_mask &= ~((ok ? 0u : 1u) << u);
                    }
                }
                base_static = ld_interior;
            };
            // channels of a slot that exist (the last chunk of a layer whose channel count is no multiple of the chunk): 0..4
            int nvalid[NSET][NU];
            auto load_slab = [&](auto set_tag) __attribute__((always_inline)) {
                constexpr int S = decltype(set_tag)::value;
                const int tbase = ld_tile * BN * p.stride - p.padL;
                if (ld_chunk == 0) setup_tile(tbase);
                const int c0 = ld_chunk * p.CC;
                if (++ld_chunk == p.nchunk) { ld_chunk = 0; ++ld_tile; }
                vmask[S] = tile_mask;
                if FC_ABL(p.ablate, 4) return;
                all_valid[S] = ld_interior;
                // the chunk (and, for interior tiles, the tile origin) goes into the buffer descriptors' scalar base, the channel of a slot's
                // four loads into the scalar offset: the lane contributes its static 32-bit descriptor only
                const size_t org = (size_t)c0 * p.Tin;
                const float* o0 = s0b + org;
                const float* o1 = s1b + org;
                if (ld_interior) { o0 += tbase; o1 += tbase; }            // tbase >= 0 here
                const __amdgpu_buffer_rsrc_t q0 = stage_rsrc(o0), q1 = stage_rsrc(o1);
                const int ch_b = 4 * p.Tin;
                if (p.cin_tail && c0 + p.CC > p.Cin) {      // last chunk runs past the real channels (uniform, rare)
                    all_valid[S] = false;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        unsigned ee = (unsigned)(rtid + 256 * u);
                        asm volatile("" : "+v"(ee));
                        const int left = p.Cin - (c0 + 4 * (int)__umulhi(ee, p.magic_slabW));
                        nvalid[S][u] = left < 0 ? 0 : (left > 4 ? 4 : left);
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            const bool okc = s4 < nvalid[S][u];
                            const unsigned off = okc ? base0[u] + (unsigned)(s4 * ch_b) : 0u;
                            v0[S][u][s4] = buf_ld1(q0, off, 0);
                            if (DUAL) v1[S][u][s4] = buf_ld1(q1, off, 0);
                        }
                    }
                    return;
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    nvalid[S][u] = 4;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        v0[S][u][s4] = buf_ld1(q0, base0[u], s4 * ch_b);
                        if (DUAL) v1[S][u][s4] = buf_ld1(q1, base0[u], s4 * ch_b);
                    }
                }
            };
            auto write_slab_t = [&](auto set_tag, char* Xd, auto use_div, auto masked) __attribute__((always_inline)) {
                constexpr int S = decltype(set_tag)::value;
                const int c0 = wr_chunk * p.CC;
                if (++wr_chunk == p.nchunk) wr_chunk = 0;
                const char* t0 = (const char*)(tab0 + c0);
                const char* t1 = (const char*)(tab1 + c0);
                float4 a0[PLAIN ? 1 : NU][2], a1[DUAL ? NU : 1][2];      // (scale, shift) of the slot's 4 channels: 32 contiguous bytes
                if (!PLAIN) {
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        a0[u][0] = *(const float4*)(t0 + cl32[u]); a0[u][1] = *(const float4*)(t0 + cl32[u] + 16);
                        if (DUAL) { a1[u][0] = *(const float4*)(t1 + cl32[u]); a1[u][1] = *(const float4*)(t1 + cl32[u] + 16); }
                    }
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    f32x4 o;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        float v = v0[S][u][s4];
                        if (!PLAIN && !FC_ABL(p.ablate, 256)) {
                            const float sc = s4 == 0 ? a0[u][0].x : s4 == 1 ? a0[u][0].z : s4 == 2 ? a0[u][1].x : a0[u][1].z;
                            const float sh = s4 == 0 ? a0[u][0].y : s4 == 1 ? a0[u][0].w : s4 == 2 ? a0[u][1].y : a0[u][1].w;
                            if (decltype(use_div)::value) v = v / divv;
                            v = fmaf(v, sc, sh);
                            if (DUAL) {
                                const float sc1 = s4 == 0 ? a1[u][0].x : s4 == 1 ? a1[u][0].z : s4 == 2 ? a1[u][1].x : a1[u][1].z;
                                const float sh1 = s4 == 0 ? a1[u][0].y : s4 == 1 ? a1[u][0].w : s4 == 2 ? a1[u][1].y : a1[u][1].w;
                                v = v + fmaf(v1[S][u][s4], sc1, sh1);
                            }
                            if (ELU) v = elu_f(v, p.alpha);
                        }
                        if (decltype(masked)::value) v = (((vmask[S] >> u) & 1u) && s4 < nvalid[S][u]) ? v : 0.f;
                        o[s4] = v;
                    }
                    *(f32x4*)(Xd + slot[u]) = o;
                }
            };
            auto write_slab = [&](auto set_tag, char* Xd) __attribute__((always_inline)) {
                constexpr int S = decltype(set_tag)::value;
                if (FC_ABL(p.ablate, 512)) { if (++wr_chunk == p.nchunk) wr_chunk = 0; return; }
                if (MODE == 1 && p.div0) write_slab_t(set_tag, Xd, std::true_type(), std::true_type());
                else if (all_valid[S]) write_slab_t(set_tag, Xd, std::false_type(), std::false_type());
                else write_slab_t(set_tag, Xd, std::false_type(), std::true_type());
            };
            using Set0 = std::integral_constant<int, 0>;
            using Set1 = std::integral_constant<int, DEEPQ ? 1 : 0>;
            int st_tile = t_begin, st_chunk = 0;
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
                __syncthreads();
                FC_STAMP(1, f, 3);
                if (++st_chunk == p.nchunk) { st_chunk = 0; flush_stats(st_tile); ++st_tile; }
                FC_STAMP(1, f, 4);
            };
            load_slab(Set0());
            write_slab(Set0(), (char*)Xs0);
            if (DEEPQ) {
                if (nitems > 1) load_slab(Set1());
                if (nitems > 2) load_slab(Set0());
            } else {
                if (nitems > 1) load_slab(Set0());
            }
            __syncthreads();                              // B0
            if (DEEPQ) {
                for (int f = 0; f < nitems; f += 2) {
                    step(f, Set1());
                    if (f + 1 < nitems) step(f + 1, Set0());
                }
            } else {
                for (int f = 0; f < nitems; ++f) step(f, Set0());
            }
            __syncthreads();                              // final
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
            if FC_ABL(p.ablate, 4) return;
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
        const bool full = !p.up_r && p.out_sT == 1 && n0 + BN <= p.Tout && m0_l + BM <= p.M && !FC_ABL(p.ablate, 2) && store_ok;
        const bool up_vec = p.up_r >= 4 && (p.up_r & 3) == 0;     // ConvTranspose1d, stride % 4 == 0: vector stores below
        const bool up_vec2 = p.up_r == 2;                         // ... stride 2: two 8-byte stores (two channels x two phases)
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
                        if (FC_ABL(p.ablate, 2) || !store_ok || up_vec || up_vec2) continue;
                        if (p.up_r) {
                            const int t = n * p.up_r + phs - p.trimL;
                            if (t >= 0 && t < p.Tfinal) rowp[t] = v;
                        } else {
                            rowp[(size_t)n * p.out_sT] = v;
                        }
                    }
                }
            }
            // Transposed convs with a stride that is a multiple of 4: the 4 accumulator rows (r & 3) of a lane are 4 consecutive GEMM rows
            // m = co * up_r + phase, i.e. 4 consecutive output samples of one channel: one 16-byte store instead of four scattered dwords
            // (round 5, profiles/r05_conv_class_ablation.txt).  The statistics above keep their (row, column tile) order: same bits as before.
            if (up_vec && !FC_ABL(p.ablate, 2) && store_ok) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int m = m0_l + wm * (TM * 32) + i * 32 + 8 * rq + 4 * hi;
                        if (m >= p.M) continue;                       // M = C_out * up_r is a multiple of the group size
                        const int co = (int)__umulhi((unsigned)m, p.magic_r), phs = m - co * p.up_r;
                        float* __restrict__ rowp = p.out + out_off + (size_t)co * p.out_sM;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int n = n0 + wn * (TN * 32) + j * 32 + l31;
                            if (n >= p.Tout) continue;
                            const float v0 = acc[i][j][4 * rq], v1 = acc[i][j][4 * rq + 1], v2 = acc[i][j][4 * rq + 2], v3 = acc[i][j][4 * rq + 3];
                            const int t = n * p.up_r + phs - p.trimL;
                            if (t >= 0 && t + 3 < p.Tfinal) {
                                *(f32x4_u*)(rowp + t) = (f32x4_u){v0, v1, v2, v3};
                            } else {
                                if (t >= 0 && t < p.Tfinal) rowp[t] = v0;
                                if (t + 1 >= 0 && t + 1 < p.Tfinal) rowp[t + 1] = v1;
                                if (t + 2 >= 0 && t + 2 < p.Tfinal) rowp[t + 2] = v2;
                                if (t + 3 >= 0 && t + 3 < p.Tfinal) rowp[t + 3] = v3;
                            }
                        }
                    }
                }
            }
            // stride 2 (the widest ConvTranspose1d of a SEANet decoder: 328 MB of output at the benchmark shape, its dword scatter was 100 of
            // 527 us): rows m, m + 1 are the two phases of channel m / 2 -- two consecutive samples -- and rows m + 2, m + 3 those of the
            // next channel: two 8-byte stores, and a wave's store covers 512 contiguous bytes instead of every other dword of 512
            if (up_vec2 && !FC_ABL(p.ablate, 2) && store_ok) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int m = m0_l + wm * (TM * 32) + i * 32 + 8 * rq + 4 * hi;
                        if (m >= p.M) continue;
                        float* __restrict__ rowp = p.out + out_off + (size_t)(m >> 1) * p.out_sM;
                        float* __restrict__ rowq = rowp + p.out_sM;
                        const bool second = m + 2 < p.M;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int n = n0 + wn * (TN * 32) + j * 32 + l31;
                            if (n >= p.Tout) continue;
                            const float v0 = acc[i][j][4 * rq], v1 = acc[i][j][4 * rq + 1], v2 = acc[i][j][4 * rq + 2], v3 = acc[i][j][4 * rq + 3];
                            const int t = 2 * n - p.trimL;
                            if (t >= 0 && t + 1 < p.Tfinal) {
                                *(f32x2_u*)(rowp + t) = (f32x2_u){v0, v1};
                                if (second) *(f32x2_u*)(rowq + t) = (f32x2_u){v2, v3};
                            } else {
                                if (t >= 0 && t < p.Tfinal) { rowp[t] = v0; if (second) rowq[t] = v2; }
                                if (t + 1 >= 0 && t + 1 < p.Tfinal) { rowp[t + 1] = v1; if (second) rowq[t + 1] = v3; }
                            }
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
        const bool stream_w = !STAGING_DMA && f + 1 < nitems && !resident && !FC_ABL(p.ablate, 16);
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
        if constexpr (QK) {
          if (!FC_ABL(p.ablate, 1)) {
            // quad-k feed: per quad (4 k-steps) ONE 16-byte LDS read per operand tile, 4 x TM x TN MFMAs; the reads of quad q + 1 are issued
            // ahead of the MFMAs of quad q (fenced, as below).  The quad count is even (a zero-weight pad quad with table offset 0 if needed);
            // the table has entries for the two quads the pipeline reads ahead.
            const f32x4* WsA = (const f32x4*)(smem + (resident ? chunk : (f & 1)) * p.Wbuf) + (hi * BM + wm * (TM * 32) + l31);
            const char* XbB = (const char*)(Xs0 + (f & 1) * XSF) + 16 * (wn * (TN * 32) + l31);
            const int* kq = kofs_i + hi;                  // this lane half's entry of a quad: table[2 q + hi]
            auto load_quad = [&](int q, int bofs, f32x4 (&a)[TM], f32x4 (&bb)[TN]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = WsA[q * 2 * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[j] = *(const f32x4*)(XbB + 4 * bofs + j * (32 * 16));
            };
            auto mfma_quad = [&](const f32x4 (&a)[TM], const f32x4 (&bb)[TN]) __attribute__((always_inline)) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s4], bb[j][s4], acc[i][j], 0, 0, 0);
            };
            f32x4 qa0[TM], qb0[TN], qa1[TM], qb1[TN];
            load_quad(0, kq[0], qa0, qb0);
            int ko = kq[2];
            for (int q = 0; q < p.nq2; q += 2) {
                load_quad(q + 1, ko, qa1, qb1);
                ko = kq[2 * q + 4];
                __builtin_amdgcn_sched_barrier(0);
                mfma_quad(qa0, qb0);
                __builtin_amdgcn_sched_barrier(0);
                if (wleft > 0) {
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)wsrc, (lds_ptr_t)wdst, 16, 0, 0);
                    wsrc += 1024; wdst += 1024; wleft -= 1024;
                }
                load_quad(q + 2, ko, qa0, qb0);
                ko = kq[2 * q + 6];
                __builtin_amdgcn_sched_barrier(0);
                // an odd quad count (k = 10 taps x 4 channels = 5 quads: encoder.model.12 of the ds640 recipe) used to multiply a whole pad
                // quad of zero weights: 16 x TM x TN MFMAs in 6 quads' worth, 17 % of that layer's matrix time
                if (!(p.nq_odd && q + 2 >= p.nq2)) mfma_quad(qa1, qb1);
                __builtin_amdgcn_sched_barrier(0);
                if (wleft > 0) {
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)wsrc, (lds_ptr_t)wdst, 16, 0, 0);
                    wsrc += 1024; wdst += 1024; wleft -= 1024;
                }
            }
          }
        } else
        if (!FC_ABL(p.ablate, 1)) {
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
            if (!FC_ABL(p.ablate, 128)) epilogue(tile, tile & 1);
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
template <int BM, int BN, int WM, int WN, int MODE, int NU, bool ROW, bool QK = false>
static hipError_t launch_conv_k(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    // the opt-in to > 64 KiB of dynamic LDS is per (kernel, device): one bit per device, set idempotently (two threads
    // racing here both set the attribute, which is harmless)
    static std::atomic<unsigned long long> attr_done{0ull};
    auto kfn = conv_mfma_kernel<BM, BN, WM, WN, MODE, NU, ROW, QK>;
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
// quad layout: slots of (channel quad, column) units per staging thread: 2 .. 5 (two-source prologues: <= 3)
static int conv_nuq_for(int totalq, int mode) {
    const int need = (totalq + 255) / 256;
    if (need <= 2) return 2;
    if (need <= 3) return 3;
    if (mode >= 3) return 0;
    return need <= 4 ? 4 : (need <= 5 ? 5 : 0);
}
// quad layout, row staging: rounds of 256 (4 channels x 4 columns) units per item (1 also covers chunks with fewer units than one round)
static int conv_rowq_rounds(int CC, int BN) {
    const int units = (CC / 4) * (BN / 4);
    static const int cw = ab_knob("FC_ROW_CW", 4);        // A / B: 1 / 2 = narrower units for chunks of exactly one round (see NU 14 / 13 in the kernel)
    if (units == 256 && cw == 1) return 14;
    if (units == 256 && cw == 2) return 13;
    if (units * 4 <= 256) return 11;          // units of 4 channels x 1 column
    if (units * 2 <= 256) return 12;          // ... x 2 columns: all four staging waves share the chunk
    return units <= 256 ? 1 : units / 256;
}

template <int BM, int BN, int WM, int WN, int MODE>
static hipError_t launch_conv_m(const ConvArgs& a, int total, dim3 grid, size_t lds, hipStream_t st) {
    if (a.row) {   // row staging: NU = rounds per item (2 / 4 / 8 specialised, 1 = run-time count)
#ifdef FC_AB_KNOBS
        // Round-4 operand layout with row staging: reachable only with FC_QUAD=0 (a tuning-build knob).  In the shipped library every chunk
        // of >= 4 channels takes the quad layout and a 2-channel chunk never qualifies for row staging (conv_row_ok), so these ~80
        // instantiations are compiled into tuning builds only (round 6: -20 % library size and build time).
        const int nr = a.CC / (4 * (64 / (BN / 4)));
        if (nr == 2) return launch_conv_k<BM, BN, WM, WN, MODE, 2, true>(a, grid, lds, st);
        if (nr == 4) return launch_conv_k<BM, BN, WM, WN, MODE, 4, true>(a, grid, lds, st);
        if constexpr (MODE < 3) {
            if (nr == 8) return launch_conv_k<BM, BN, WM, WN, MODE, 8, true>(a, grid, lds, st);
        }
        return launch_conv_k<BM, BN, WM, WN, MODE, 1, true>(a, grid, lds, st);
#else
        return hipErrorInvalidValue;
#endif
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

// quad-layout instantiations (their own translation units: conv_tileq_*.hip).  Round 6: the 40 instantiations of a tile shape are spread over
// THREE translation units by prologue mode (conv_tileq_<tile>.hip: the launcher + modes 5, 3, 4; _m01.hip: modes 0, 1; _m2.hip: mode 2) --
// the 128 x 128 unit alone was 95 s of a 156 s cold build.  FC_TIMELINE builds keep one unit per tile (the stamps live in a per-unit symbol).
template <int BM, int BN, int WM, int WN, int MODE>
hipError_t launch_conv_mq(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    if constexpr (MODE == 5) {      // DMA-staged slab: NU = 16-byte pieces per lane of a staging wave (256 per round)
        const int nj = ((a.CC / 4) * a.rowStride + 255) / 256;
        if (nj == 1) return launch_conv_k<BM, BN, WM, WN, 5, 1, false, true>(a, grid, lds, st);
        if (nj == 2) return launch_conv_k<BM, BN, WM, WN, 5, 2, false, true>(a, grid, lds, st);
        if (nj == 3) return launch_conv_k<BM, BN, WM, WN, 5, 3, false, true>(a, grid, lds, st);
        if (nj == 4) return launch_conv_k<BM, BN, WM, WN, 5, 4, false, true>(a, grid, lds, st);
        if (nj == 5) return launch_conv_k<BM, BN, WM, WN, 5, 5, false, true>(a, grid, lds, st);
        if (nj == 6) return launch_conv_k<BM, BN, WM, WN, 5, 6, false, true>(a, grid, lds, st);
        return hipErrorInvalidValue;
    } else {
    if (a.row) {
        const int nr = conv_rowq_rounds(a.CC, BN);
        if (nr == 1) return launch_conv_k<BM, BN, WM, WN, MODE, 1, true, true>(a, grid, lds, st);
        if (nr == 12) return launch_conv_k<BM, BN, WM, WN, MODE, 12, true, true>(a, grid, lds, st);
        if (nr == 11) return launch_conv_k<BM, BN, WM, WN, MODE, 11, true, true>(a, grid, lds, st);
#ifdef FC_AB_KNOBS
        if (nr == 13) return launch_conv_k<BM, BN, WM, WN, MODE, 13, true, true>(a, grid, lds, st);
        if (nr == 14) return launch_conv_k<BM, BN, WM, WN, MODE, 14, true, true>(a, grid, lds, st);
#endif
        if constexpr (MODE < 3) {
            if (nr == 2) return launch_conv_k<BM, BN, WM, WN, MODE, 2, true, true>(a, grid, lds, st);
        }
        return hipErrorInvalidValue;
    }
    const int nu = conv_nuq_for((a.CC / 4) * a.slabW, MODE);
    if (nu == 2) return launch_conv_k<BM, BN, WM, WN, MODE, 2, false, true>(a, grid, lds, st);
    if (nu == 3) return launch_conv_k<BM, BN, WM, WN, MODE, 3, false, true>(a, grid, lds, st);
    if constexpr (MODE < 3) {
        if (nu == 4) return launch_conv_k<BM, BN, WM, WN, MODE, 4, false, true>(a, grid, lds, st);
        if (nu == 5) return launch_conv_k<BM, BN, WM, WN, MODE, 5, false, true>(a, grid, lds, st);
    }
    return hipErrorInvalidValue;
    }
}

#ifndef FC_TIMELINE
#define FC_CONVQ_ELSEWHERE(BM, BN, WM, WN, MODE) extern template hipError_t launch_conv_mq<BM, BN, WM, WN, MODE>(const ConvArgs&, dim3, size_t, hipStream_t);
#define FC_CONVQ_HERE(BM, BN, WM, WN, MODE) template hipError_t launch_conv_mq<BM, BN, WM, WN, MODE>(const ConvArgs&, dim3, size_t, hipStream_t);
#else
#define FC_CONVQ_ELSEWHERE(BM, BN, WM, WN, MODE)
#define FC_CONVQ_HERE(BM, BN, WM, WN, MODE)
#endif

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
template <int BM, int BN, int WM, int WN>
hipError_t launch_conv_tile_q(const ConvLaunch& c, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    if (a.xq_Tp > 0) return launch_conv_mq<BM, BN, WM, WN, 5>(a, grid, lds, st);
    if (c.s1.ptr) return c.elu ? launch_conv_mq<BM, BN, WM, WN, 4>(a, grid, lds, st) : launch_conv_mq<BM, BN, WM, WN, 3>(a, grid, lds, st);
    if (c.s0.aff || c.s0.div || c.elu) return c.elu ? launch_conv_mq<BM, BN, WM, WN, 2>(a, grid, lds, st)
                                                    : launch_conv_mq<BM, BN, WM, WN, 1>(a, grid, lds, st);
    return launch_conv_mq<BM, BN, WM, WN, 0>(a, grid, lds, st);
}

// which template instantiation launch_conv() will pick (profiling labels)
}  // namespace fc
