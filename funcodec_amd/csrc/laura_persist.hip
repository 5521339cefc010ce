// LauraTTS decoding step as ONE persistent launch (round 4).
//
// The step form of the codec language model (laura_kernels.hip: 49 weight-streaming GEMVs + 12 cache attentions per step, 8..16 token
// vectors) is a chain of DEPENDENT all-to-all edges: every kernel of the round-3 chain cost ~8 us of which < 1 us was work (launch
// boundary, first memory round trip, ramp-up / drain), 498 us per step for 272 MB of traffic.  Here the whole step (12 blocks + the
// output layer) is one launch of G <= 256 co-resident workgroups (one per CU, 8 waves); what used to be a kernel boundary is a counter
// the producers of a phase arrive on and only the consumers of that phase wait for:
//
//   phase 5l + 0  QKV   3d/16 row tiles   x_l -> LayerNorm -> q (edge buffer), k / v (KV caches, write-through)
//   phase 5l + 1  ATT   B * H * NS units  one (utterance, head, key range): flash-decoding partial (o, max, sum)
//   phase 5l + 2  OUT   d/16 tiles        combine the partials while staging, linear_out, + x_l            -> xm_l
//   phase 5l + 3  FF1   ff/16 tiles       LayerNorm(xm_l), w_1, activation                                  -> h_l
//   phase 5l + 4  FF2   KS2 x d/16 units  (k slice, row tile) of w_2 h_l; slice 0 adds xm_l and the bias      -> KS2 partial x_{l+1}
//   phase 5 NL    DEC   ceil(V/16) tiles  after_norm, output layer                                          -> logits (next kernel: sampler)
//
// * The weights of a unit do not depend on data: while a workgroup computes its current GEMV unit, waves 1..7 request the 16-row tile of
//   its NEXT GEMV unit by LDS DMA (`global_load_lds_dwordx4`: fragment order, 1 KiB per wave instruction, straight into the other half
//   of a double-buffered LDS tile, no registers), so the weight stream (155 MB per step, the only real HBM traffic) runs under the
//   hand-offs and waits instead of behind them.  (A register-resident prefetch was built first: hipcc spilled the prefetched tile or
//   waited for it right behind the loads -- the 248 registers the attention unit takes leave no room to carry it across.)  Wave 0 issues
//   no weight loads: it polls, stores and arrives, so its `s_waitcnt vmcnt(0)` drains stores only and its polls do not queue behind a tile.
// * w_2 (K = ff) is split into k slices of ~d columns (step_persist_ksplit): every GEMV unit contracts over ~d values (short staging and
//   MFMA chains on the critical path, one LDS tile size); the consumers of a block's output add the slices while staging.
// * Hand-off = the persistent LSTM's (kernels.hip): results are stored WRITE-THROUGH (relaxed agent-scope atomic stores -> sc1), the storing
//   wave drains its stores (s_waitcnt vmcnt(0)) and adds 1 to one of the phase's 16 arrival counters (separate cache lines, monotonic
//   targets = launch number x producers); consumers poll with 16 lanes, then read the edge buffer with PLAIN loads: every edge buffer
//   is written once per launch and never read before its phase is complete, so no cache can hold a stale copy (copies of the previous
//   launch die at the kernel boundary, like the buffers of the kernel chain).  Spins are bounded: on a timeout the error word is set,
//   every workgroup runs to completion and the host falls back to the kernel chain.
// * Arithmetic: the chain's (fp32-input 16x16x4 MFMA chains over K, K split over 8 waves summed in wave order; two-pass LayerNorm; exact
//   softmax with the flash-decoding split), same formulas as gemv_kernel / attn_step_kernel; the split of K and of the key ranges differs,
//   so results agree with the chain to fp32 rounding, not bit for bit.
#include "laura_kernels.h"

#include <atomic>
#include <cstdlib>

namespace fc {
namespace laura {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
// Pointers that reach the kernel through a table in memory (StepLayer) are generic to the compiler: it would emit flat_load (LDS
// aperture check, counted in lgkmcnt as well).  Everything here lives in global memory: say so.
typedef const __attribute__((address_space(1))) float* gfp;
typedef const __attribute__((address_space(1))) f32x4* gv4p;
typedef const __attribute__((address_space(1))) f32x4u* gv4up;
#define FC_G(p) ((gfp)(p))
#define FC_G4(p) ((gv4p)(p))

namespace {

// profiling builds only (FC_BUILD_DEFINES="FC_PERSIST_ABL=<mask>", results are garbage): 1 no LayerNorm arithmetic, 2 reciprocal / rsqrt
// approximations instead of the precise division and square root in it
#ifndef FC_PERSIST_ABL
#define FC_PERSIST_ABL 0
#endif

constexpr int kThreads = 512, kWaves = 8;
constexpr int kCntStride = 32;            // words between two arrival counters (128 bytes)
constexpr int kCntPerPhase = 16;

// sum / max over the 64 lanes of a wave: 16-lane rows by DPP butterflies (quad swaps, half-row and row mirrors), then the four row
// results through readlane -- ~20 VALU instructions; a __shfl_xor tree is 6 ds_bpermute round trips through the LDS crossbar
#define FC_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))
#define FC_LANE(v, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l))
__device__ __forceinline__ float wave_sum_p(float v) {
    v += FC_DPP(v, 0xB1);
    v += FC_DPP(v, 0x4E);
    v += FC_DPP(v, 0x141);
    v += FC_DPP(v, 0x140);
    return ((FC_LANE(v, 0) + FC_LANE(v, 16)) + FC_LANE(v, 32)) + FC_LANE(v, 48);
}
__device__ __forceinline__ float wave_max_p(float v) {
    v = fmaxf(v, FC_DPP(v, 0xB1));
    v = fmaxf(v, FC_DPP(v, 0x4E));
    v = fmaxf(v, FC_DPP(v, 0x141));
    v = fmaxf(v, FC_DPP(v, 0x140));
    return fmaxf(fmaxf(FC_LANE(v, 0), FC_LANE(v, 16)), fmaxf(FC_LANE(v, 32), FC_LANE(v, 48)));
}
__device__ __forceinline__ float act_p(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v / (1.f + expf(-v));
    return v;
}
__device__ __forceinline__ void store_wt(float* p, float v) {       // write-through store (global_store_dword ... sc1)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_wt4(float* p, f32x4 v) {      // 16-byte write-through store: one fabric write instead of four
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// Workgroup barrier for LDS hand-offs that leaves global memory operations in flight.  __syncthreads() is a workgroup-scope fence +
// s_barrier: hipcc drains vmcnt(0) in front of it, i.e. every barrier would wait for the weight tile this wave has just requested (the
// first two builds: +1.1 us per unit).  LDS traffic is ordered by lgkmcnt; global hand-offs have their own explicit drains (arrive).
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Item { int ph, u; };

// tuning aid: thread 0 of a workgroup stamps s_memtime (shader clock) at the edges of a unit's life; slot = (workgroup, n-th unit)
struct Trace {
    unsigned long long* p;
    __device__ __forceinline__ void stamp(int j) const { if (p && threadIdx.x == 0) p[j] = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void id(int ph, int u) const { if (p && threadIdx.x == 0) p[7] = ((unsigned long long)(unsigned)ph << 32) | (unsigned)u; }
};

// kinds: 0 QKV, 1 ATT, 2 OUT, 3 FF1, 4 FF2, 5 DEC
__device__ __forceinline__ int phase_kind(const StepPersistArgs& a, int ph) { return ph == 5 * a.NL ? 5 : ph % 5; }
__device__ __forceinline__ int kind_units(const StepPersistArgs& a, int k) {
    switch (k) {
        case 0: return 3 * a.d / 16;
        case 1: return a.B * a.H * a.NS;
        case 2: return a.d / 16;
        case 3: return a.ff / 16;
        case 4: return a.KS2 * (a.d / 16);            // w_2 split over K: unit = (k slice, row tile), the consumers add the slices
        default: return (a.V + 15) / 16;
    }
}
// first workgroup of a kind: QKV tiles from 0, OUT and FF2 tiles behind them, FF1 tiles behind those (a workgroup then holds at most
// ~2 weight tiles of a block); attention units and the output layer from 0
__device__ __forceinline__ int kind_off(const StepPersistArgs& a, int k) {
    const int tq = 3 * a.d / 16, to = a.d / 16;
    switch (k) {
        case 2: case 4: return tq % a.G;
        case 3: return (tq + to) % a.G;
        default: return 0;
    }
}
__device__ __forceinline__ bool next_item(const StepPersistArgs& a, int wg, Item& it) {
    const int nph = 5 * a.NL + 1;
    int ph = it.ph, u = it.u + a.G;
    for (;;) {
        if (ph >= 0 && u < kind_units(a, phase_kind(a, ph))) { it.ph = ph; it.u = u; return true; }
        if (++ph >= nph) { it.ph = -1; it.u = 0; return false; }
        u = wg - kind_off(a, phase_kind(a, ph));
        if (u < 0) u += a.G;
    }
}

// ---- arrival counters ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned* phase_counter(const StepPersistArgs& a, int ph, int c) {
    return a.sync + ((size_t)ph * kCntPerPhase + c) * kCntStride;
}
__device__ __forceinline__ unsigned* error_word(const StepPersistArgs& a) {
    return a.sync + (size_t)(5 * a.NL + 1) * kCntPerPhase * kCntStride;
}
// the storing wave (wave 0): its write-through stores are in memory once vmcnt drains; then one arrival
__device__ __forceinline__ void arrive(const StepPersistArgs& a, int ph, int unit) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) __hip_atomic_fetch_add(phase_counter(a, ph, unit & (kCntPerPhase - 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every producer of phase `ph` of THIS launch has arrived (wave 0 polls, one counter per lane); ends with a workgroup barrier
__device__ __forceinline__ void wait_phase(const StepPersistArgs& a, int ph, unsigned seq) {
    if (ph >= 0 && threadIdx.x < 64) {
        const int c = threadIdx.x & (kCntPerPhase - 1);
        const int units = kind_units(a, phase_kind(a, ph));
        const unsigned target = seq * (unsigned)(units > c ? (units - c + kCntPerPhase - 1) / kCntPerPhase : 0);
        const unsigned* mine = phase_counter(a, ph, c);
        unsigned spins = 0;
        for (;;) {
            const unsigned v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v >= target)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0u) {
                if (__hip_atomic_load(error_word(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (spins > (1u << 21)) { __hip_atomic_store(error_word(a), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
    }
    wg_barrier();
}

// ---- weight tiles ------------------------------------------------------------------------------------------------------------------
struct TileDesc {
    gfp wf, bias;             // wf: first fragment chunk of this unit's row tile
    int K, N, nch, cpw;       // K, nch = K / 16: this unit's share of the contraction; cpw = 16-wide k chunks per wave
    int tile, ks;             // row tile; k slice (w_2 only, else 0)
};
__device__ __forceinline__ int cpw_for(int nch) { return (nch + kWaves - 1) / kWaves; }
__device__ __forceinline__ TileDesc tile_desc(const StepPersistArgs& a, Item it) {
    TileDesc t{nullptr, nullptr, 0, 0, 0, 0, 0, 0};
    if (it.ph < 0) return t;
    const int k = phase_kind(a, it.ph);
    t.tile = it.u;
    int nch_row = a.d / 16;                           // chunks of a whole weight row
    if (k == 5) { t.wf = FC_G(a.wdec); t.bias = FC_G(a.bdec); t.K = a.d; t.N = a.V; }
    else {
        const StepLayer& L = a.layers[it.ph / 5];
        if (k == 0) { t.wf = FC_G(L.wqkv); t.bias = FC_G(L.bqkv); t.K = a.d; t.N = 3 * a.d; }
        else if (k == 2) { t.wf = FC_G(L.wout); t.bias = FC_G(L.bout); t.K = a.d; t.N = a.d; }
        else if (k == 3) { t.wf = FC_G(L.wff1); t.bias = FC_G(L.bff1); t.K = a.d; t.N = a.ff; }
        else {
            const int tiles = a.d / 16;
            t.ks = it.u / tiles; t.tile = it.u - t.ks * tiles;
            t.wf = FC_G(L.wff2); t.bias = FC_G(L.bff2); t.K = a.ff / a.KS2; t.N = a.d;
            nch_row = a.ff / 16;
        }
    }
    t.nch = t.K / 16;
    t.cpw = cpw_for(t.nch);
    t.wf += ((size_t)t.tile * nch_row + (size_t)t.ks * t.nch) * 256;
    return t;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Request a unit's weight tile (nch chunks of 1 KiB, fragment order) and its 16 bias values into LDS: chunks striped over waves 1..7, one
// `global_load_lds_dwordx4` per chunk (lane i's 16 bytes land at dst + 16 i), the bias by wave 1 (lanes 0..15: one dword each).
__device__ __forceinline__ void request_tile(const TileDesc& t, float* Wd, float* bias_d, int wv, int lane) {
    if (wv == 0 || t.wf == nullptr) return;
    for (int c = wv - 1; c < t.nch; c += kWaves - 1)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(t.wf + (size_t)c * 256 + lane * 4), (lds_ptr_t)(Wd + c * 256), 16, 0, 0);
    if (wv == 1) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(t.bias + t.tile * 16 + (lane & 15)), (lds_ptr_t)bias_d, 4, 0, 0);   // bias arrays are padded to whole row tiles
}

__device__ __forceinline__ f32x4 mfma_tile(const float* Wc, const TileDesc& t, const float* Xs, int XS, int B, int wv, int lane) {
    const int g = lane >> 4, r16 = lane & 15;
    const float* xb = Xs + (r16 < B ? r16 : B) * XS + 4 * g;      // rows >= B read the zero row
    const float* wb = Wc + lane * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int c0 = wv * t.cpw;
    int c1 = c0 + t.cpw;
    c1 = c1 < t.nch ? c1 : t.nch;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const f32x4 wq = *(const f32x4*)(wb + c * 256);
        const f32x4 xv = *(const f32x4*)(xb + c * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[j], xv[j], acc, 0, 0, 0);
    }
    return acc;
}

// ---- edge buffers --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float* edge_q(const StepPersistArgs& a, int l) { return a.edge + (size_t)l * a.edge_stride + a.o_q; }
__device__ __forceinline__ float* edge_ap(const StepPersistArgs& a, int l) { return a.edge + (size_t)l * a.edge_stride + a.o_ap; }
__device__ __forceinline__ float* edge_xm(const StepPersistArgs& a, int l) { return a.edge + (size_t)l * a.edge_stride + a.o_xm; }
__device__ __forceinline__ float* edge_hb(const StepPersistArgs& a, int l) { return a.edge + (size_t)l * a.edge_stride + a.o_hb; }
__device__ __forceinline__ float* edge_xo(const StepPersistArgs& a, int l) { return a.edge + (size_t)l * a.edge_stride + a.o_xo; }
// the input vector of block l = the step's input (l = 0) or the SUM of the KS2 k-slice outputs of the previous block's w_2
__device__ __forceinline__ const float* layer_in(const StepPersistArgs& a, int l) { return l == 0 ? a.xs : edge_xo(a, l - 1); }
__device__ __forceinline__ int layer_in_parts(const StepPersistArgs& a, int l) { return l == 0 ? 1 : a.KS2; }

// ---- staging of a GEMV unit's input into LDS: Xs [B + 1][XS] (row B = zeros), optional LayerNorm over K ------------------------------
// src: `parts` buffers of [16][ld] floats (part stride 16 * ld) that are added up; columns [col0, col0 + K) of every row.
// Every load of a thread is issued before the first use (batches of 4 elements x PM parts, clamped addresses): a load -> LDS store loop
// pays one memory round trip per trip, and a fresh line of another XCD's write-through store is ~1 us away (measured: 7.6 us per QKV
// unit in the first version of this kernel, of which 3.6 us were a LayerNorm that walked LDS three times per row).
template <int PM>
__device__ __forceinline__ void stage_rows(const float* src, int ld, int col0, int parts, int K, int B, int XS, float* Xs, const float* gamma,
                                           const float* beta, float eps, float* gb, Trace tr) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int K4 = K >> 2, NE = B * K4;
    for (int k = tid; k < K4; k += kThreads) *(f32x4*)(Xs + B * XS + 4 * k) = (f32x4){0.f, 0.f, 0.f, 0.f};        // the zero row
    if (gamma) {       // K <= 1024: one 16-byte piece of gamma or beta per thread
        const int i = tid < 2 * K4 ? tid : 0;
        const f32x4 gv = *FC_G4(FC_G(i < K4 ? gamma : beta) + 4 * (i < K4 ? i : i - K4));
        if (tid < 2 * K4) *(f32x4*)(gb + 4 * i) = gv;
    }
    for (int base = 0; base < NE; base += 4 * kThreads) {
        f32x4 v[4][PM];
        int dst[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = base + i * kThreads + tid, ec = e < NE ? e : NE - 1;
            const int b = ec / K4, k4 = ec - b * K4;
            dst[i] = e < NE ? b * XS + 4 * k4 : -1;
            gfp sp = FC_G(src) + (size_t)b * ld + col0 + 4 * k4;
#pragma unroll
            for (int q = 0; q < PM; ++q) v[i][q] = *FC_G4(sp + (size_t)(q < parts ? q : 0) * 16 * ld);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 sum = v[i][0];
#pragma unroll
            for (int q = 1; q < PM; ++q) if (q < parts) sum += v[i][q];
            if (dst[i] >= 0) *(f32x4*)(Xs + dst[i]) = sum;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the weight tile (LDS DMA) has landed too
    tr.stamp(5);
    wg_barrier();
    tr.stamp(6);
    if (gamma) {       // two-pass LayerNorm, one wave per row, the row in registers (<= 4 pieces of 16 bytes per lane)
        for (int b = w; b < B && !(FC_PERSIST_ABL & 1); b += kWaves) {
            float* xr = Xs + b * XS;
            f32x4 x[4];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = lane + 64 * i;
                x[i] = j < K4 ? *(const f32x4*)(xr + 4 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
                s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
            }
            f32x4 gm[4], bt[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = lane + 64 * i < K4 ? lane + 64 * i : 0;
                gm[i] = *(const f32x4*)(gb + 4 * j);
                bt[i] = *(const f32x4*)(gb + K + 4 * j);
            }
            const float mean = (FC_PERSIST_ABL & 2) ? wave_sum_p(s) * __builtin_amdgcn_rcpf((float)K) : wave_sum_p(s) / (float)K;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (lane + 64 * i < K4) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const float dv = x[i][c] - mean; q += dv * dv; }
                }
            const float rstd = (FC_PERSIST_ABL & 2) ? __builtin_amdgcn_rsqf(wave_sum_p(q) * __builtin_amdgcn_rcpf((float)K) + eps)
                                                    : 1.f / sqrtf(wave_sum_p(q) / (float)K + eps);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = lane + 64 * i;
                if (j < K4) *(f32x4*)(xr + 4 * j) = (x[i] - mean) * rstd * gm[i] + bt[i];
            }
        }
        wg_barrier();
    }
}

// x[b][h * DK + dd] = combination of the NS key-range partials of head h (flash-decoding): o_s (unnormalised), running max m_s, sum l_s;
// partial row = [o[DK] | m | l | pad pad] (16-byte aligned).  Everything a thread needs is requested at once: its (b, h) pair's m / l
// values and the o pieces of its <= 2 output quads per batch.
__device__ __forceinline__ void stage_partials(const float* apart, int B, int H, int DK, int NS, int XS, float* Xs, float* wn, Trace tr) {
    const int tid = threadIdx.x, K = H * DK, PS = DK + 4, DK4 = DK >> 2;
    const int NE = B * H * DK4;                       // output quads
    for (int k = tid * 4; k < K; k += kThreads * 4) *(f32x4*)(Xs + B * XS + k) = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mv[8], lv[8];
    {
        const int e = tid < B * H ? tid : 0;
        gfp pp = FC_G(apart) + (size_t)e * NS * PS + DK;
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) {
            const int sc = sp < NS ? sp : NS - 1;
            mv[sp] = pp[sc * PS];
            lv[sp] = pp[sc * PS + 1];
        }
    }
    bool first = true;
    for (int base = 0; base < NE; base += 2 * kThreads) {
        f32x4 ov[2][8];
        int bh[2], dst[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = base + i * kThreads + tid, ec = e < NE ? e : NE - 1;
            bh[i] = ec / DK4;
            const int q4 = ec - bh[i] * DK4, b = bh[i] / H, h = bh[i] - b * H;
            dst[i] = e < NE ? b * XS + h * DK + 4 * q4 : -1;
            gfp pp = FC_G(apart) + (size_t)bh[i] * NS * PS + 4 * q4;
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) ov[i][sp] = *FC_G4(pp + (size_t)(sp < NS ? sp : NS - 1) * PS);
        }
        if (first) {       // normalised weights exp(m_s - M) / L of every (b, h), once
            first = false;
            if (tid < B * H) {
                float M = -INFINITY;
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) M = fmaxf(M, sp < NS ? mv[sp] : -INFINITY);
                float L = 0.f, wg[8];
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) {
                    wg[sp] = sp < NS ? expf(mv[sp] - M) : 0.f;
                    L = fmaf(lv[sp], wg[sp], L);
                }
                const float inv = 1.f / L;
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) wn[tid * 8 + sp] = wg[sp] * inv;
            }
            wg_barrier();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* wq = wn + bh[i] * 8;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) o += ov[i][sp] * wq[sp];          // weights of ranges >= NS are 0
            if (dst[i] >= 0) *(f32x4*)(Xs + dst[i]) = o;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the weight tile (LDS DMA) has landed too
    tr.stamp(5);
    wg_barrier();
}

// ---- attention unit: one (utterance, head) against ONE key range of the KV cache; the two halves of the workgroup take half of the
// range each (attn_step_kernel's pass structure per half) and meet in LDS -----------------------------------------------------------------
template <int DK>
__device__ __forceinline__ void att_unit(const StepPersistArgs& a, int l, int u, float* lds) {
    constexpr int DG = 8, DPG = DK / DG;
    constexpr int NQ = DK / 4, NG = 256 / NQ;
    constexpr int CH = 128, NV = CH / NG;
    // LDS carve (floats)
    float* part = lds;                                // [2][DG][CH]
    float* sc = part + 2 * DG * CH;                   // [2][CH]
    float* cred = sc + 2 * CH;                        // [2][NG][DK]
    float* qu = cred + 2 * NG * DK;                   // [DK]
    float* qv = qu + DK;                              // [DK]
    float* wred = qv + DK;                            // [2][4]
    float* ml = wred + 8;                             // [2][2]: running max, sum of each half
    const StepLayer& L = a.layers[l];
    const int tid = threadIdx.x, hf = tid >> 8, t = tid & 255, lane = tid & 63, wl = (tid >> 6) & 3;
    const int d = a.H * DK;
    const int b = u / (a.H * a.NS), rem = u - b * a.H * a.NS, h = rem / a.NS, sp = rem - h * a.NS;
    const int p = a.pos[b], n = p + 1;
    const int chunk = ((n + a.NS - 1) / a.NS + 3) & ~3;
    const int k0 = sp * chunk;
    const int k1 = k0 + chunk < n ? k0 + chunk : n;
    const int len = k1 > k0 ? k1 - k0 : 0;
    const int hc = ((len + 1) / 2 + 3) & ~3;          // keys per half, whole quads
    const int h0 = k0 + hf * hc;
    const int h1 = h0 + hc < k1 ? h0 + hc : k1;
    const int npass = (hc + CH - 1) / CH;             // the same for both halves (workgroup barriers inside the pass loop)
    float* out = edge_ap(a, l) + ((size_t)(b * a.H + h) * a.NS + sp) * (DK + 4);
    if (tid < DK) {
        const float q = edge_q(a, l)[(size_t)b * d + h * DK + tid];
        qu[tid] = q + FC_G(L.bu)[h * DK + tid];
        qv[tid] = q + FC_G(L.bv)[h * DK + tid];
    }
    const int dg = t >> 5, ql = t & 31;
    const int dq = t % NQ, jg = t / NQ;
    gfp kb = FC_G(a.kc) + (size_t)l * a.B * d * a.Tcap + ((size_t)b * d + h * DK + dg * DPG) * a.Tcap;
    gfp pb = FC_G(L.ptab) + (size_t)(h * DK + dg * DPG) * a.PR + (a.R - 1) + p;
    gfp vb = FC_G(a.vc) + (size_t)l * a.B * d * a.Tcap + (size_t)b * a.Tcap * d + h * DK + 4 * dq;
    float* partH = part + hf * DG * CH;
    float* scH = sc + hf * CH;
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o_acc = {0.f, 0.f, 0.f, 0.f};
    const int cbase = h0 < n ? h0 : 0;                // a valid key for clamped loads of an empty half
    for (int ps = 0; ps < npass; ++ps) {
        const int c0 = h0 + ps * CH;
        int cn = h1 - c0;
        cn = cn < 0 ? 0 : (cn > CH ? CH : cn);
        const int cs = cn > 0 ? c0 : cbase;           // where clamped loads point
        const bool sq = 4 * ql < cn;
        const int jq = sq ? c0 + 4 * ql : cs;
        f32x4 kv[DPG], pv[DPG];
#pragma unroll
        for (int dd = 0; dd < DPG; ++dd) {
            kv[dd] = *FC_G4(kb + (size_t)dd * a.Tcap + jq);
            pv[dd] = *(gv4up)(pb + (size_t)dd * a.PR - jq - 3);
        }
        f32x4 vv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int jj = jg + i * NG;
            vv[i] = *FC_G4(vb + (size_t)(jj < cn ? c0 + jj : cs) * d);
        }
        wg_barrier();                              // qu / qv visible (first pass); LDS of the previous pass free
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dd = 0; dd < DPG; ++dd) {
            const float uu = qu[dg * DPG + dd], vq = qv[dg * DPG + dd];
            acc[0] = fmaf(uu, kv[dd][0], fmaf(vq, pv[dd][3], acc[0]));
            acc[1] = fmaf(uu, kv[dd][1], fmaf(vq, pv[dd][2], acc[1]));
            acc[2] = fmaf(uu, kv[dd][2], fmaf(vq, pv[dd][1], acc[2]));
            acc[3] = fmaf(uu, kv[dd][3], fmaf(vq, pv[dd][0], acc[3]));
        }
        *(f32x4*)&partH[dg * CH + 4 * ql] = acc;
        wg_barrier();
        const float scale = 1.f / sqrtf((float)DK);
        float s = -INFINITY;
        if (t < cn) {
            float tt = 0.f;
#pragma unroll
            for (int gq = 0; gq < DG; ++gq) tt += partH[gq * CH + t];
            s = tt * scale;
        }
        float m = wave_max_p(s);
        if (lane == 0) wred[hf * 4 + wl] = m;
        wg_barrier();
        m = fmaxf(fmaxf(wred[hf * 4], wred[hf * 4 + 1]), fmaxf(wred[hf * 4 + 2], wred[hf * 4 + 3]));
        const float m_new = fmaxf(m_run, m);
        const float e = t < cn ? expf(s - m_new) : 0.f;
        if (t < CH) scH[t] = e;
        float ls = wave_sum_p(e);
        wg_barrier();
        if (lane == 0) wred[hf * 4 + wl] = ls;
        wg_barrier();
        if (cn > 0) {                                 // an empty pass of this half leaves its running state alone (m_new may be -inf)
            const float corr = expf(m_run - m_new);
            l_run = l_run * corr + (wred[hf * 4] + wred[hf * 4 + 1] + wred[hf * 4 + 2] + wred[hf * 4 + 3]);
            m_run = m_new;
            o_acc *= corr;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int jj = jg + i * NG;
                if (jj < cn) o_acc += scH[jj] * vv[i];
            }
        }
    }
    *(f32x4*)&cred[(hf * NG + jg) * DK + 4 * dq] = o_acc;
    if (t == 0) { ml[hf * 2] = m_run; ml[hf * 2 + 1] = l_run; }
    wg_barrier();
    if (tid < 64) {                                   // wave 0 stores (and arrives afterwards)
        const float m0 = ml[0], l0 = ml[1], m1 = ml[2], l1 = ml[3];
        const float M = fmaxf(m0, m1);
        const float w0 = m0 == -INFINITY ? 0.f : expf(m0 - M), w1 = m1 == -INFINITY ? 0.f : expf(m1 - M);
        if (tid < DK) {
            float o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) { o0 += cred[gq * DK + tid]; o1 += cred[(NG + gq) * DK + tid]; }
            store_wt(out + tid, o0 * w0 + o1 * w1);
        }
        if (tid == 0) { store_wt(out + DK, M); store_wt(out + DK + 1, l0 * w0 + l1 * w1); }
    }
}

// ---- one GEMV unit of the step ----------------------------------------------------------------------------------------------------------
// LDS map (floats): weight tile 0 | weight tile 1 | bias row 0 | bias row 1 | work region (GEMV staging or attention scratch).  Addresses
// are computed from the kernel's shared array, never stored in a pointer table (a table makes them generic: flat loads).
__device__ __forceinline__ float* lds_tile(float* lds, const StepPersistArgs& a, int par) { return lds + par * a.wtile; }
__device__ __forceinline__ float* lds_bias(float* lds, const StepPersistArgs& a, int par) { return lds + 2 * a.wtile + par * 64; }
__device__ __forceinline__ float* lds_work(float* lds, const StepPersistArgs& a) { return lds + 2 * a.wtile + 128; }

// `par`: which LDS tile holds THIS unit's weights (requested while the previous GEMV unit computed); the next unit's go to the other one
__device__ __forceinline__ void gemv_item(const StepPersistArgs& a, Item it, int par, Item nxt, unsigned seq, float* lds, Trace tr) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const int k = phase_kind(a, it.ph), l = it.ph / 5;
    const TileDesc t = tile_desc(a, it);
    const TileDesc tn = tile_desc(a, nxt);
    const int B = a.B, K = t.K, XS = K + 4;
    float* Xs = lds_work(lds, a);
    float* gb = Xs + (size_t)(B + 1) * XS;            // gamma | beta (2 * d floats), then the partial weights of the OUT prologue
    float* red = gb + 2 * a.d + 8 * B * a.H;          // [kWaves][64][4]
    const int eb = r16 < B ? r16 : 0, n0 = t.tile * 16 + 4 * g;
    const int epos = a.pos[eb];
    tr.id(it.ph, it.u);
    tr.stamp(0);
    wait_phase(a, it.ph - 1, seq);
    tr.stamp(1);
    // Epilogue operands, requested together with the staging loads.  An edge buffer may only be read once its phase is complete (a line
    // read earlier could be served stale later: a workgroup without an attention unit reaches OUT(l) straight from its own FF2(l - 1) tile,
    // before the other tiles of x_l exist), so these loads sit BEHIND the wait, and the ADDRESS is selected, never the load: OUT adds the
    // block's input (= the sum of the previous w_2's k slices), FF2 adds xm; every other kind reads the step's input vector and ignores it.
    const int nn = n0 < a.d ? n0 : 0;
    gfp rsrc = FC_G(k == 2 ? layer_in(a, l) : (k == 4 ? edge_xm(a, l) : a.xs)) + (size_t)eb * a.d + nn;
    const int rparts = k == 2 ? layer_in_parts(a, l) : 1;
    f32x4 eold = *FC_G4(rsrc);
    for (int q = 1; q < rparts; ++q) eold += *FC_G4(rsrc + (size_t)q * 16 * a.d);
    const StepLayer& L = a.layers[k == 5 ? 0 : l];
    switch (k) {
        case 0: stage_rows<4>(layer_in(a, l), a.d, 0, layer_in_parts(a, l), K, B, XS, Xs, L.n1g, L.n1b, 1e-12f, gb, tr); break;
        case 2: stage_partials(edge_ap(a, l), B, a.H, a.DK, a.NS, XS, Xs, gb + 2 * a.d, tr); break;
        case 3: stage_rows<1>(edge_xm(a, l), a.d, 0, 1, K, B, XS, Xs, L.n2g, L.n2b, 1e-12f, gb, tr); break;
        case 4: stage_rows<1>(edge_hb(a, l), a.ff, t.ks * K, 1, K, B, XS, Xs, nullptr, nullptr, 0.f, gb, tr); break;
        default: stage_rows<4>(layer_in(a, a.NL), a.d, 0, layer_in_parts(a, a.NL), K, B, XS, Xs, a.ag, a.ab, 1e-12f, gb, tr); break;
    }
    tr.stamp(2);
    const f32x4 acc = mfma_tile(lds_tile(lds, a, par), t, Xs, XS, B, wv, lane);
    *(f32x4*)(red + (wv * 64 + lane) * 4) = acc;
    // The next GEMV unit's tile streams into the other LDS tile under this unit's hand-off and the next wait.  Requested BEHIND this
    // wave's last LDS access of the unit: hipcc drains vmcnt before the first LDS access that follows an LDS DMA (it cannot tell the
    // tiles apart) -- ahead of the MFMA loop that put the tile's HBM round trip (1.1 us) into every unit.
    request_tile(tn, lds_tile(lds, a, par ^ 1), lds_bias(lds, a, par ^ 1), wv, lane);
    wg_barrier();
    tr.stamp(3);
    if (wv == 0) {
        f32x4 s = *(const f32x4*)(red + lane * 4);
#pragma unroll
        for (int x = 1; x < kWaves; ++x) s += *(const f32x4*)(red + (x * 64 + lane) * 4);
        const f32x4 pb = *(const f32x4*)(lds_bias(lds, a, par) + 4 * g);
        const int b = r16;
        if (t.ks == 0) s += pb;                       // bias (and FF2's residual) with the first k slice only
        if (b < B) {
            if (k == 0) {
                const int d = a.d;                    // a row tile lies inside q, k or v (d % 16 == 0)
                if (n0 < d) store_wt4(edge_q(a, l) + (size_t)b * d + n0, s);
                else if (n0 < 2 * d) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) store_wt(a.kc + (size_t)l * B * d * a.Tcap + ((size_t)b * d + (n0 + r - d)) * a.Tcap + epos, s[r]);
                } else store_wt4(a.vc + (size_t)l * B * d * a.Tcap + ((size_t)b * a.Tcap + epos) * d + (n0 - 2 * d), s);
            } else if (k == 2) store_wt4(edge_xm(a, l) + (size_t)b * a.d + n0, eold + s);
            else if (k == 3) {
                f32x4 hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = act_p(s[r], a.act);
                store_wt4(edge_hb(a, l) + (size_t)b * a.ff + n0, hv);
            } else if (k == 4) store_wt4(edge_xo(a, l) + ((size_t)t.ks * 16 + b) * a.d + n0, t.ks == 0 ? eold + s : s);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < t.N) a.logits[(size_t)b * a.V + n0 + r] = s[r];       // read by the next kernel (the sampler)
            }
        }
        arrive(a, it.ph, it.u);
        tr.stamp(4);
    }
}

__device__ __forceinline__ void att_item(const StepPersistArgs& a, Item it, unsigned seq, float* work, Trace tr) {
    tr.id(it.ph, it.u);
    tr.stamp(0);
    wait_phase(a, it.ph - 1, seq);
    tr.stamp(1);
    if (a.DK == 64) att_unit<64>(a, it.ph / 5, it.u, work);
    else att_unit<32>(a, it.ph / 5, it.u, work);
    tr.stamp(3);
    if (threadIdx.x < 64) arrive(a, it.ph, it.u);
    tr.stamp(4);
}

__device__ __forceinline__ bool same_item(Item x, Item y) { return x.ph == y.ph && x.u == y.u; }
// the first GEMV unit at or behind `it` in this workgroup's schedule (attention units carry no weights); ph = -1: none
__device__ __forceinline__ Item gemv_from(const StepPersistArgs& a, int wg, Item it) {
    while (it.ph >= 0 && phase_kind(a, it.ph) == 1) next_item(a, wg, it);
    return it;
}

}  // namespace

__global__ __launch_bounds__(kThreads, 2) void step_persist_kernel(StepPersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned seq = *a.seq;
    if (a.test_timeout && wg == 0 && tid == 0) __hip_atomic_store(error_word(a), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Item it{-1, -a.G};
    next_item(a, wg, it);                 // first unit of this workgroup (phase order)
    if (it.ph < 0) return;
    Item g = gemv_from(a, wg, it);
    int par = 0, nth = 0;
    request_tile(tile_desc(a, g), lds_tile(lds, a, 0), lds_bias(lds, a, 0), wv, lane);
    auto slot = [&]() { Trace t{a.trace ? a.trace + ((size_t)wg * 64 + (nth < 63 ? nth : 63)) * 8 : nullptr}; ++nth; return t; };
    for (;;) {
        while (it.ph >= 0 && !same_item(it, g)) { att_item(a, it, seq, lds_work(lds, a), slot()); next_item(a, wg, it); }     // attention units ahead of g
        if (g.ph < 0) break;
        next_item(a, wg, it);
        const Item g2 = gemv_from(a, wg, it);
        gemv_item(a, g, par, g2, seq, lds, slot());
        g = g2;
        par ^= 1;
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------------------
size_t step_persist_sync_words(int NL) { return (size_t)(5 * NL + 1) * kCntPerPhase * kCntStride + 64; }

int step_persist_ksplit(int d, int ff) {
    // w_2 (K = ff) is split into k slices of about d columns each (at most 4), one unit per (slice, row tile): every GEMV unit of the step
    // then contracts over ~d values -- shorter staging and MFMA chains on the critical path, a smaller register set for the prefetched tile
    int ks = ff / d;
    ks = ks < 1 ? 1 : (ks > 4 ? 4 : ks);
    while (ks > 1 && ff % (16 * ks)) --ks;
    return ks;
}

namespace {
int unit_kmax(int d, int ff) {
    const int k2 = ff / step_persist_ksplit(d, ff);
    return d > k2 ? d : k2;
}
}  // namespace

size_t step_persist_lds_bytes(int B, int d, int ff, int H, int DK) {
    const int Kmax = unit_kmax(d, ff);
    const size_t tiles = ((size_t)2 * Kmax * 16 + 128) * sizeof(float);                  // two weight tiles + two bias rows
    const size_t gemv = ((size_t)(B + 1) * (Kmax + 4) + 2 * (size_t)d + 8 * (size_t)B * H + (size_t)kWaves * 256) * sizeof(float);
    const int NQ = DK / 4, NG = 256 / NQ;
    const size_t att = ((size_t)2 * 8 * 128 + 2 * 128 + (size_t)2 * NG * DK + 2 * DK + 8 + 4) * sizeof(float);
    return tiles + (gemv > att ? gemv : att) + 64;
}

bool step_persist_supported(int B, int d, int ff, int H, int DK, int V, int NS) {
    if (B < 1 || B > 16 || d % 16 || ff % 16 || d > 1024 || (DK != 32 && DK != 64) || H * DK != d || V < 1) return false;
    if (NS < 1 || NS > 8) return false;
    return step_persist_lds_bytes(B, d, ff, H, DK) <= 160 * 1024;
}

int step_persist_grid(int device, int d, int ff) {
    // one workgroup per CU, all co-resident (the hand-offs need every producer running); the occupancy query is the check
    (void)d; (void)ff;
    const void* fn = (const void*)step_persist_kernel;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kThreads, 8 * 1024) != hipSuccess || per_cu < 1) return 0;
    int G = prop.multiProcessorCount;
    return G > 256 ? 256 : G;
}

hipError_t launch_step_persist(const StepPersistArgs& a, hipStream_t st) {
    const size_t lds = step_persist_lds_bytes(a.B, a.d, a.ff, a.H, a.DK);
    const void* fn = (const void*)step_persist_kernel;
    StepPersistArgs args = a;
    args.wtile = unit_kmax(a.d, a.ff) * 16;
    void* params[] = {&args};
    return hipLaunchKernel(fn, dim3(a.G), dim3(kThreads), params, lds, st);
}

}  // namespace laura
}  // namespace fc
