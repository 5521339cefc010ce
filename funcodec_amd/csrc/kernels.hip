// gfx950 (MI355X / CDNA4) kernels of the FunCodec encode/decode hot path.
//
// Everything here is fp32.  Dense contractions (conv-as-implicit-GEMM, LSTM recurrence, RVQ distance
// matrix) run on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32): on gfx950
// those are bit-for-bit an fmaf chain in k order at the fp32 vector peak rate, i.e. no precision is
// traded.  Wavefront = 64 lanes throughout.
#include "conv_kernel.h"

namespace fc {

// per-tile launchers of the implicit-GEMM conv kernel: instantiated in conv_tile_*.hip
#define FC_CONV_TILE(BM, BN, WM, WN) \
    extern template hipError_t launch_conv_tile<BM, BN, WM, WN>(const ConvLaunch&, const ConvArgs&, dim3, size_t, hipStream_t);
FC_CONV_TILE(128, 128, 2, 2)
FC_CONV_TILE(64, 256, 1, 4)
FC_CONV_TILE(32, 256, 1, 4)
FC_CONV_TILE(32, 128, 1, 4)
#undef FC_CONV_TILE
// ... and their quad-layout twins (conv_tileq_*.hip)
#define FC_CONV_TILE_Q(BM, BN, WM, WN) \
    extern template hipError_t launch_conv_tile_q<BM, BN, WM, WN>(const ConvLaunch&, const ConvArgs&, dim3, size_t, hipStream_t);
FC_CONV_TILE_Q(128, 128, 2, 2)
FC_CONV_TILE_Q(64, 256, 1, 4)
FC_CONV_TILE_Q(32, 256, 1, 4)
FC_CONV_TILE_Q(32, 128, 1, 4)
#undef FC_CONV_TILE_Q

// QUAD-K operand layout (round 5, tests/micro/conv_loop_feed_b128.hip: 0.95 - 0.98 of the fp32 MFMA peak from the LDS feed against 0.81 /
// 0.89 for one ds_read_b32 per operand tile per k-step): chunks of a multiple of 8 channels keep both operands so that a lane's values
// of FOUR consecutive k-steps are one 16-byte piece.  FC_QUAD=0 restores the round-4 layout (A / B aid).
bool conv_quad(int CC) {
    static const int quad_env = ab_knob("FC_QUAD", 1);
    return quad_env != 0 && CC >= 4 && CC % 4 == 0;
}
// A chunk's k indices are grouped into HALF-QUADS (tap kk, channel quad g4) in tap-major order h = kk * CC/4 + g4; quad q = half-quads 2q
// (lanes 0-31 of the MFMA operands) and 2q + 1 (lanes 32-63).  Quads per chunk, rounded up to an even count (the matrix loop runs two
// quads per trip; pad half-quads have zero weights and table offset 0)
static int conv_nquads(int k, int CC) { return (((k * (CC / 4) + 1) / 2) + 1) & ~1; }
static int conv_koff_len(int k, int CC) { return conv_quad(CC) ? ((2 * conv_nquads(k, CC) + 3) & ~3) + 8 : (((k * CC / 2) + 3) & ~3) + 8; }
// float index of W[m = mm of the tile][chunk channel cl][tap kk] inside a packed chunk image
//   round-4 layout: [kk][cl][BM]                      (k-step = two consecutive channels)
//   quad layout:    [quad = h / 2][hi = h & 1][BM][s = cl & 3],  h = kk * CC/4 + cl/4     (k-step s of a quad = channel s of its two half-quads)
size_t conv_pack_index(int k, int CC, int BM, int kk, int cl, int mm) {
    (void)k;
    if (!conv_quad(CC)) return ((size_t)kk * CC + cl) * BM + mm;
    const int h = kk * (CC / 4) + cl / 4, s4 = cl & 3;
    return (((size_t)(h >> 1) * 2 + (h & 1)) * BM + mm) * 4 + s4;
}

// copy of the timeline stamps of a profiling build (zeros otherwise): [role][item][slot]
hipError_t debug_timeline_128(unsigned long long* dst);   // conv_tile_128x128.hip (the profiling build stamps that tile shape)
hipError_t debug_timeline_128q(unsigned long long* dst);  // conv_tileq_128x128.hip
hipError_t debug_timeline(unsigned long long* dst) { return conv_quad(4) ? debug_timeline_128q(dst) : debug_timeline_128(dst); }

static ConvArgs make_args(const ConvLaunch& c) {
    ConvArgs a;
    a.src0 = c.s0.ptr; a.aff0 = c.s0.aff; a.div0 = c.s0.div;
    a.src1 = c.s1.ptr; a.aff1 = c.s1.aff;
    a.wt = c.wt; a.bias = c.bias; a.out = c.out; a.partials = c.partials;
    a.out_sB = c.out_sB; a.out_sM = c.out_sM; a.out_sT = c.out_sT;
    a.Fo = c.Fo < 1 ? 1 : c.Fo;
    a.affC = c.affC > 0 ? c.affC : c.Cin;
    a.in_sB0 = c.in_sB0 ? c.in_sB0 : (long long)c.Cin * c.Tin;
    a.in_sB1 = c.in_sB1;
    a.out_sF = c.out_sF;
    a.part_sB0 = c.part_sB0 ? c.part_sB0 : (long long)a.Fo * conv_nblk(c);
    a.store_lo = c.store_lo; a.store_hi = c.store_hi;
    a.B = c.B; a.Cin = c.Cin; a.Tin = c.Tin; a.M = c.M; a.Tout = c.Tout;
    a.k = c.k; a.stride = c.stride; a.padL = c.padL; a.padR = c.padR; a.pad_zero = c.pad_zero;
    a.dil = c.dil;
    const int maxpad = c.padL > c.padR ? c.padL : c.padR;
    a.Leff = c.Tin > maxpad ? c.Tin : maxpad + 1;
    a.up_r = c.up_r; a.trimL = c.trimL; a.Tfinal = c.Tfinal;
    a.magic_r = c.up_r ? (unsigned)((0x100000000ull + c.up_r - 1) / (unsigned long long)c.up_r) : 0u;
    a.elu = c.elu; a.alpha = c.alpha;
    a.CC = c.CC; a.nchunk = c.nchunk; a.Kc = c.k * c.CC;
    a.Wbuf = conv_wbuf_floats(c.k, c.CC, c.BM);
    a.slabW = (c.BN - 1) * c.stride + (c.k - 1) * c.dil + 1;
    a.PL = ceil_div(a.slabW, c.stride);
    a.rowStride = a.PL * c.stride;
    a.row = c.row;
    if (c.row) { a.rowStride = (a.slabW + 3) & ~3; a.PL = a.rowStride; }   // 16-byte aligned rows
    // floats per slab buffer: row staging = the image; element staging = whole 256-element slots (the last float is the dummy
    // slot of lanes without an element)
    a.xsf = c.row ? c.CC * a.rowStride + 4 : ceil_div(c.CC * a.rowStride, 256) * 256 + 4;
    a.xq_Tp = c.xq_Tp;
    if (conv_quad(c.CC) && c.xq_Tp > 0)
        a.xsf = ceil_div(c.CC * a.rowStride, 1024) * 1024 + 4;      // DMA-staged slabs (MODE 5): whole 256-piece rounds.  Only there: a
                                                                    // register-staged plain launch must not use more LDS than its layer was budgeted
    a.magic_slabW = (unsigned)(0x100000000ull / (unsigned long long)a.slabW) + 1u;
    a.koff = c.koff;
    a.koff_n = conv_koff_len(c.k, c.CC);
    a.quad = conv_quad(c.CC) ? 1 : 0;
    a.nq2 = conv_nquads(c.k, c.CC);
    a.nq_odd = (conv_quad(c.CC) && ((((c.k * (c.CC / 4)) + 1) / 2) & 1)) ? 1 : 0;
    // quad element staging: units of (4 channels, slab column)
    a.magic_slabW = (unsigned)(0x100000000ull / (unsigned long long)a.slabW) + 1u;
    a.cin_tail = (c.Cin % c.CC) != 0;
    static const int ablate = ab_knob("FC_ABLATE", 0);
    a.ablate = ablate;
    return a;
}

// B-operand LDS offset (floats) of every k-step of a chunk: k-step ks covers kidx = 2*ks + {0,1} = kk*CC + 2*c2 + {0,1}
// -> offset = 2*c2*rowStride + (kk % stride)*PL + kk / stride.  Padded to a multiple of 4 entries.
std::vector<int> conv_koff_table(int k, int stride, int dil, int CC, int BN, int row) {
    const int slabW = (BN - 1) * stride + (k - 1) * dil + 1;
    int PL = ceil_div(slabW, stride), rowStride = PL * stride;
    if (row) { rowStride = (slabW + 3) & ~3; PL = rowStride; }
    if (conv_quad(CC)) {
        // entry 2 q + hi = FLOAT offset of half-quad h = 2 q + hi: its channel plane (planes are [g4][rowStride columns][4 channels]) and the
        // column of its tap
        const int n4 = CC / 4;
        std::vector<int> t(conv_koff_len(k, CC), 0);
        for (int h = 0; h < k * n4; ++h) {
            const int kk = h / n4, g4 = h % n4, pos = kk * dil;
            t[h] = (g4 * rowStride + (pos % stride) * PL + pos / stride) * 4;
        }
        return t;
    }
    const int nks = k * CC / 2, half_cc = CC / 2;
    std::vector<int> t(conv_koff_len(k, CC), 0);
    for (int ks = 0; ks < nks; ++ks) {
        const int kk = ks / half_cc, c2 = ks % half_cc, pos = kk * dil;   // slab column of tap kk relative to the output column
        t[ks] = 2 * c2 * rowStride + (pos % stride) * PL + pos / stride;
    }
    return t;
}

// K of a chunk is zero-padded to a multiple of 4 k-steps (8 k values): the main loop has no tail
int conv_wbuf_floats(int k, int CC, int BM) {
    if (conv_quad(CC)) return ((conv_nquads(k, CC) * 8 * BM + 1023) / 1024) * 1024;
    return ((((k * CC + 7) & ~7) * BM + 1023) / 1024) * 1024;
}

bool conv_cout1_ok(const ConvLaunch& c);
// outputs per workgroup of the few-output FMA kernels: the LDS form (1024) on short rows / 2-D layers, the streaming form (992) else
static int conv_fewout_tile(const ConvLaunch& c) { return (c.Fo > 1 || c.Tout < 8192) ? 1024 : 992; }
bool conv_fewout_rows(const ConvLaunch& c) { return conv_fewout_tile(c) == 1024; }
static bool conv_fewout_plain(const ConvLaunch& c) {        // rows form, one materialised source, no prologue arithmetic: all-DMA staging
    static const int plain_env = ab_knob("FC_FEWOUT_PLAIN", 1);     // 0: the general rows kernel (A / B aid)
    static const int ablate_env = ab_knob("FC_ABLATE", 0);
    return conv_fewout_rows(c) && plain_env && !c.s0.aff && !c.s1.ptr && !c.elu && !ablate_env;
}
void conv_fewout_name(const ConvLaunch& c, char* buf, size_t n) {
    if (conv_fewout_plain(c)) snprintf(buf, n, "conv_fewout_rows_plain_kernel<%d, %d>", c.k, c.M);
    else snprintf(buf, n, "%s<%d, %s, %d>", conv_fewout_rows(c) ? "conv_fewout_rows_kernel" : "conv_cout1_kernel", c.k, c.s1.ptr ? "true" : "false", c.M);
}
int conv_nblk(const ConvLaunch& c) {
    if (conv_cout1_ok(c)) return ceil_div(c.Tout, conv_fewout_tile(c));      // few-output FMA kernels: one partial per workgroup
    return ceil_div(c.Tout, c.BN) * ceil_div(c.M, c.BM);
}

// Row staging (stride 1): rows per round = 4 waves x (64 lanes / (BN/4) lanes per row); the kernel holds at most 8
// rounds of one source (4 of two) in registers.
bool conv_row_ok(int k, int stride, int dil, int CC, int BM, int BN, int Cin, bool dual) {
    if (stride != 1 || (BN != 128 && BN != 256)) return false;
    if (conv_quad(CC)) {
        // quad layout: units of (4 channels x 4 columns), 256 per round: one or two whole rounds (two-source prologues: one -- a unit holds
        // 16 values per source, more rounds spill) or less than one; one tail unit (4 channels x 1 column) per thread at most
        if (Cin % CC != 0) return false;
        const int units = (CC / 4) * (BN / 4);
        if (units > 256 && units % 256 != 0) return false;
        if (units / 256 > (dual ? 1 : 2)) return false;
        return (CC / 4) * (k - 1) * dil <= 256 && (size_t)conv_wbuf_floats(k, CC, BM) <= 8192;
    }
    const int rpr = 4 * (64 / (BN / 4));
    if (CC % rpr != 0 || Cin % CC != 0) return false;
    if (CC / rpr > (dual ? 4 : 8)) return false;
    return CC * (k - 1) * dil <= 256 && (size_t)conv_wbuf_floats(k, CC, BM) <= 8192;
}

size_t conv_lds_bytes_for(int k, int stride, int dil, int CC, int BM, int BN, int Cin, int ntab, int row, int dma_rounds) {
    const int slabW = (BN - 1) * stride + (k - 1) * dil + 1;
    const int rowStride = row ? ((slabW + 3) & ~3) : ceil_div(slabW, stride) * stride;
    const int img = CC * rowStride;
    int xs = row ? img + 4 : ceil_div(img, 256) * 256 + 4;   // make_args(): xsf
    // dma_rounds: the slab buffer holds whole DMA rounds (MODE 5).  -1 (the planner's budget): assumed for every layer without tables, so
    // that the budget is an upper bound of what any launch of the layer takes
    if (conv_quad(CC) && (dma_rounds > 0 || (dma_rounds < 0 && ntab == 0))) xs = ceil_div(img, 1024) * 1024 + 4;
    const size_t koff_bytes = (size_t)conv_koff_len(k, CC) * sizeof(int);
    return (size_t)(2 * conv_wbuf_floats(k, CC, BM) + 2 * xs) * sizeof(float) + (size_t)((Cin + 1) & ~1) * 8 * ntab + koff_bytes +
           (size_t)BM * sizeof(float) + 2 * 256 * 8;
}

// Every tiling runs two 512-thread workgroups per CU (128 VGPRs per wave).  Three workgroups of the 32-row tiles at 80
// VGPRs forced the 8-element staging variant and therefore 4x smaller K chunks: measured slower (per-item overheads).
int conv_wgs_per_cu(int) { return 2; }
bool conv_slab_fits(int k, int stride, int dil, int CC, int BN, int BM, bool dual) {
    const int slabW = (BN - 1) * stride + (k - 1) * dil + 1;
    const int img = CC * ceil_div(slabW, stride) * stride;
    if (conv_quad(CC)) return (CC / 4) * slabW <= (dual ? 3 : 5) * 256;    // quad slots of 4 channels: 3 two-source, 5 otherwise
    if (dual) return img <= NU_DUAL * 256;             // two-source prologue: at most NU_DUAL slots (two values + two table entries each)
    return img <= SLAB_PER_THREAD * 256;
}

size_t conv_lds_bytes(const ConvLaunch& c) {
    const int ntab = c.s1.ptr ? 2 : ((c.s0.aff || c.s0.div || c.elu) ? 1 : 0);
    return conv_lds_bytes_for(c.k, c.stride, c.dil, c.CC, c.BM, c.BN, c.Cin, ntab, c.row, c.xq_Tp > 0 ? 1 : 0);
}

void conv_variant(const ConvLaunch& c, int* mode, int* nu, int* row) {
    const ConvArgs a = make_args(c);
    *row = a.row;
    if (c.s1.ptr) *mode = c.elu ? 4 : 3;
    else if (c.s0.aff || c.s0.div || c.elu) *mode = c.elu ? 2 : 1;
    else *mode = 0;
    if (a.quad && a.xq_Tp > 0) {
        *mode = 5; *nu = ((a.CC / 4) * a.rowStride + 255) / 256; *row = 2;
        return;
    }
    if (a.quad) {
        *nu = a.row ? conv_rowq_rounds(a.CC, c.BN) : conv_nuq_for((a.CC / 4) * a.slabW, *mode);
        *row |= 2;                                         // bit 1: quad layout (profile label)
    } else if (a.row) {
        const int nr = a.CC / (4 * (64 / (c.BN / 4)));
        *nu = (nr == 2 || nr == 4 || (nr == 8 && *mode < 3)) ? nr : 1;
    } else {
        *nu = conv_nu_for(a.CC * a.rowStride, *mode);
    }
}

static hipError_t launch_conv_cout1(const ConvLaunch& c, hipStream_t st);
hipError_t launch_conv(const ConvLaunch& c, hipStream_t st) {
    if (conv_cout1_ok(c)) return launch_conv_cout1(c, st);
    const ConvArgs a = make_args(c);
    const size_t lds = conv_lds_bytes(c);
    if (lds > 160 * 1024 || (c.CC & 1)) return hipErrorInvalidValue;
    // the staging waves address a chunk's rows with 32-bit byte offsets from a per-chunk base
    if ((unsigned long long)c.CC * (unsigned long long)c.Tin * 4ull >= (1ull << 32)) return hipErrorInvalidValue;
    if (c.xq_Tp > 0) {
        if (!a.quad || c.s1.ptr || c.s0.aff || c.s0.div || c.elu || (c.Fo > 1) || c.Cin % c.CC != 0 ||
            !conv_xq_ok(c.Cin, c.CC, c.k, c.stride, c.dil, c.BM, c.BN, c.row) || c.xq_Tp != c.padL + c.Tin + c.padR)
            return hipErrorInvalidValue;
    } else if (c.row) {
        if (!conv_row_ok(c.k, c.stride, c.dil, c.CC, c.BM, c.BN, c.Cin, c.s1.ptr != nullptr) || c.s0.div) return hipErrorInvalidValue;
    } else if (a.quad) {
        if (!conv_slab_fits(c.k, c.stride, c.dil, c.CC, c.BN, c.BM, c.s1.ptr != nullptr)) return hipErrorInvalidValue;
    } else if (c.CC * (ceil_div((c.BN - 1) * c.stride + (c.k - 1) * c.dil + 1, c.stride) * c.stride) > SLAB_PER_THREAD * 256) {
        return hipErrorInvalidValue;
    }
    // one resident wave of workgroups (2 per CU): each takes a contiguous range of N tiles
    const int ntiles = ceil_div(c.Tout, c.BN), mtiles = ceil_div(c.M, c.BM);
    static const int target_env = deploy_switch("FC_TARGET_WGS", 0);
    const int target_wgs = target_env ? target_env : 256 * conv_wgs_per_cu(c.BM);
    int G = target_wgs / (mtiles * c.B);
    if (G < 1) G = 1;
    if (G > ntiles) G = ntiles;
    if (!target_env) {
        // One resident wave of workgroups walks max ceil(ntiles / G) tiles each; when that split is uneven (decoder.model.6.convtr of the
        // ds640 recipe: 16 column tiles over 3 workgroups per (utterance, M tile) = 6 / 5 / 5, i.e. 6 tile times for 5.33 tiles of work), MORE and
        // shorter ranges dispatched over several rounds balance better: rounds x (tiles per workgroup + ~6 % start-up per workgroup).  Taken
        // only when the model predicts >= 5 % (round 6; results do not depend on G: FC_TARGET_WGS test).
        auto cost = [&](int g) {
            const long long wgs = (long long)g * mtiles * c.B;
            return ((wgs + target_wgs - 1) / target_wgs) * (100ll * ceil_div(ntiles, g) + 6);
        };
        int best = G;
        const int g_hi = ntiles < 8 * G + 16 ? ntiles : 8 * G + 16;      // a bounded host-side search: the candidates that matter are small multiples
        for (int g = G + 1; g <= g_hi; ++g)
            if (cost(g) < cost(best)) best = g;
        if (cost(best) * 100 <= cost(G) * 95) G = best;
    }
    dim3 grid(G, mtiles, c.B);
    if (a.quad) {
        if (c.BM == 128 && c.BN == 128) return launch_conv_tile_q<128, 128, 2, 2>(c, a, grid, lds, st);
        if (c.BM == 64 && c.BN == 256) return launch_conv_tile_q<64, 256, 1, 4>(c, a, grid, lds, st);
        if (c.BM == 32 && c.BN == 256) return launch_conv_tile_q<32, 256, 1, 4>(c, a, grid, lds, st);
        if (c.BM == 32 && c.BN == 128) return launch_conv_tile_q<32, 128, 1, 4>(c, a, grid, lds, st);
        return hipErrorInvalidValue;
    }
    if (c.BM == 128 && c.BN == 128) return launch_conv_tile<128, 128, 2, 2>(c, a, grid, lds, st);
    if (c.BM == 64 && c.BN == 256) return launch_conv_tile<64, 256, 1, 4>(c, a, grid, lds, st);
    if (c.BM == 32 && c.BN == 256) return launch_conv_tile<32, 256, 1, 4>(c, a, grid, lds, st);
    if (c.BM == 32 && c.BN == 128) return launch_conv_tile<32, 128, 1, 4>(c, a, grid, lds, st);
    return hipErrorInvalidValue;
}

// =================================================================================================
// 1b. Stride-1 conv with 1..4 output channels (the decoder's last layer: 32 -> 1, k = 7; FreqCodec's last Conv2d: 7 rows x 32
//     channels -> 3): a 32-row MFMA tile would do 8..32x the necessary matrix work -> plain FMAs, weights through scalar loads.
//     Same epilogue contract as the MFMA kernel: raw output + fp64 (sum, sum of squares) partial per workgroup (992 outputs).
// =================================================================================================
struct Cout1Args {
    const float *src0, *aff0, *src1, *aff1;    // [B][Cin][T], per-(b,c) affine or null
    const float* w;                            // [M][Cin][k]
    const float* bias;                         // [M] device
    float* out;                                // [B][M][T]
    double* partials;                          // [B][ntiles][2] or null
    int Cin, T, k, padL, Leff, elu;
    float alpha;
    // two-level batch (2-D convs over frequency-major activations, see ConvArgs): b = breal * Fo + fo
    int Fo, affC;
    long long in_sB0, in_sB1, out_sB, out_sF, out_sM, part_sB0;
    int ablate;          // FC_ABLATE env (profiling aid): 1 no FMAs, 2 no stores, 4 no global loads, 8 no prologue math
};
// LDS form for SHORT rows (the 2-D nets: ~1 000 samples per frequency row, many input channels): one workgroup = 1024 outputs of one
// (virtual) utterance, the prologue'd input staged 8 channels at a time into LDS ([8][1032]), every thread owns 4 consecutive outputs and
// reads its 4 + k - 1 slab columns with three 16-byte LDS loads per channel; the next chunk's global loads are in flight while the
// current one is multiplied.  (Measured on FreqCodec's last Conv2d, 224 -> 3 channels: 2.1 ms per launch against 3.5 ms for the
// streaming form below, which wins on the long 1-D rows: 187 vs 225 us.)
constexpr int C1L_TN = 1024, C1L_CH = 8, C1L_ROW = 1032;

template <int K, bool DUAL, int MO>
__global__ __launch_bounds__(256) void conv_fewout_rows_kernel(const Cout1Args p) {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    __shared__ __attribute__((aligned(16))) float Xs[C1L_CH][C1L_ROW];
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int breal = p.Fo > 1 ? b / p.Fo : b, fo = p.Fo > 1 ? b - breal * p.Fo : 0;
    const int t0 = tile * C1L_TN, tbase = t0 - p.padL;
    const size_t in_off = (size_t)breal * p.in_sB0 + (size_t)fo * p.in_sB1;
    const float* s0 = p.src0 + in_off;
    const float* s1 = DUAL ? p.src1 + in_off : s0;
    const bool wrap = p.affC != p.Cin;         // virtual channel = frequency row * affC + real channel
    const float2* a0 = p.aff0 ? (const float2*)p.aff0 + (size_t)breal * p.affC : nullptr;
    const float2* a1 = (DUAL && p.aff1) ? (const float2*)p.aff1 + (size_t)breal * p.affC : nullptr;
    // source index (reflect padding, conv.py:82-99) and validity of this thread's 4 main columns + (threads 0..7) one tail column.
    // A thread whose 4 main columns are plain in-range samples takes one 16-byte load per row instead of four gathers.
    const int g0 = tbase + 4 * tid;
    const bool vec_ok = g0 >= 0 && g0 + 3 < p.T;
    int esrc[5]; unsigned emask = 0;
    {
        const int refl = 2 * (p.Leff - 1);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int col = j < 4 ? 4 * tid + j : 1024 + tid;
            const int g = tbase + col;
            int src = g < 0 ? -g : g;
            src = src >= p.Leff ? refl - src : src;
            const bool ok = (j < 4 || tid < 8) && g >= -p.padL && src >= 0 && src < p.T;
            esrc[j] = ok ? src : 0;
            emask |= (ok ? 1u : 0u) << j;
        }
    }
    auto act = [&](float v, float w, float2 A, float2 A1) __attribute__((always_inline)) {
        if FC_ABL(p.ablate, 8) return v;
        v = fmaf(v, A.x, A.y);
        if (DUAL) v = v + fmaf(w, A1.x, A1.y);
        if (p.elu) v = elu_f(v, p.alpha);
        return v;
    };
    float acc[MO][4];
#pragma unroll
    for (int m = 0; m < MO; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
    // raw values of the chunk being loaded: the loads of chunk c+1 are in flight while chunk c is multiplied out of LDS
    f32x4 raw0[C1L_CH], raw1[DUAL ? C1L_CH : 1];
    float rt0[C1L_CH], rt1[DUAL ? C1L_CH : 1];
    // straight-line loads (clamped addresses, no branch around them); threads at a padded edge re-gather when they stage
    const int gsafe = g0 < 0 ? 0 : (g0 + 3 < p.T ? g0 : (p.T >= 4 ? p.T - 4 : 0));
    auto load_chunk = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < C1L_CH; ++r) {
            const int c = c0 + r < p.Cin ? c0 + r : p.Cin - 1;
            const float* r0 = s0 + (size_t)c * p.T;
            const float* r1 = s1 + (size_t)c * p.T;
            raw0[r] = *(const f32x4u*)(r0 + gsafe);
            if (DUAL) raw1[r] = *(const f32x4u*)(r1 + gsafe);
            rt0[r] = r0[esrc[4]];                                        // tail column (threads 0..7 use it; index 0 elsewhere)
            if (DUAL) rt1[r] = r1[esrc[4]];
        }
    };
    const unsigned vmask = vec_ok ? 0xFu | (emask & 0x10u) : emask;
    load_chunk(0);
    for (int c0 = 0; c0 < p.Cin; c0 += C1L_CH) {
        // ---- activate + stage 8 channels
#pragma unroll
        for (int r = 0; r < C1L_CH; ++r) {
            const int c = c0 + r;
            const bool cok = c < p.Cin;
            const int ca = !cok ? 0 : wrap ? c % p.affC : c;
            const float2 A = a0 ? a0[ca] : make_float2(1.f, 0.f);
            const float2 A1 = a1 ? a1[ca] : make_float2(1.f, 0.f);
            f32x4 w0 = raw0[r], w1 = DUAL ? raw1[r] : raw0[r];
            if (!vec_ok && cok) {                                        // padded edge: gather by index
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w0[j] = s0[(size_t)c * p.T + esrc[j]];
                    if (DUAL) w1[j] = s1[(size_t)c * p.T + esrc[j]];
                }
            }
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = (cok && ((vmask >> j) & 1u)) ? act(w0[j], w1[j], A, A1) : 0.f;
            *(f32x4*)&Xs[r][4 * tid] = v;
            if (tid < 8) Xs[r][1024 + tid] = (cok && ((vmask >> 4) & 1u)) ? act(rt0[r], DUAL ? rt1[r] : 0.f, A, A1) : 0.f;
        }
        __syncthreads();
        load_chunk(c0 + C1L_CH);                                         // past the last channel: clamped re-read, dropped
        // ---- 4 outputs per thread and output channel: out[m][n] += sum_c sum_kk w[m][c][kk] * x[c][n + kk]
#pragma unroll
        for (int r = 0; r < C1L_CH; ++r) {
            const int c = c0 + r;
            if (c >= p.Cin || FC_ABL(p.ablate, 1)) break;                   // uniform
            const f32x4 q0 = *(const f32x4*)&Xs[r][4 * tid];
            const f32x4 q1 = *(const f32x4*)&Xs[r][4 * tid + 4];
            const f32x4 q2 = *(const f32x4*)&Xs[r][4 * tid + 8];
            const float x[12] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3], q2[0], q2[1], q2[2], q2[3]};
#pragma unroll
            for (int m = 0; m < MO; ++m) {
                const float* wr = p.w + ((size_t)m * p.Cin + c) * K;   // uniform address: scalar loads
#pragma unroll
                for (int kk = 0; kk < K; ++kk) {
                    const float wv = wr[kk];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][j] = fmaf(wv, x[j + kk], acc[m][j]);
                }
            }
        }
        __syncthreads();
    }
    // ---- epilogue: bias, store, statistics of the valid outputs
    float s1v = 0.f, s2v = 0.f;
#pragma unroll
    for (int m = 0; m < MO; ++m) {
        float o[4];
        const float bm = p.bias[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = acc[m][j] + bm;
            const int n = t0 + 4 * tid + j;
            if (n < p.T) { s1v += o[j]; s2v = fmaf(o[j], o[j], s2v); }
        }
        float* orow = p.out + (size_t)breal * p.out_sB + (size_t)fo * p.out_sF + (size_t)m * p.out_sM + t0 + 4 * tid;
        if (t0 + 4 * tid + 3 < p.T) *(f32x4u*)orow = (f32x4){o[0], o[1], o[2], o[3]};
        else
#pragma unroll
            for (int j = 0; j < 4; ++j) if (t0 + 4 * tid + j < p.T) orow[j] = o[j];
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
            const size_t slot = ((size_t)breal * p.part_sB0 + (size_t)fo * gridDim.x + tile) * 2;
            p.partials[slot] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
            p.partials[slot + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        }
    }
}

// PLAIN-input form of the rows kernel (round 4): the source is one already materialised tensor (no affine, no ELU, one source -- the 2-D
// nets' last Conv2d, whose input `combine2d` activates once).  The profile of the kernel above on that layer (2.3 ms per 32-utterance call,
// 0.21 of the fp32 peak) is its STAGING, not its 672 FMAs per 8 channels: 1 500 instructions and ~100 waits to gather / select / activate / store
// 8 channel rows into LDS, two barriers per chunk, nothing in flight during the arithmetic.  Here every staged value goes HBM -> LDS by DMA
// (no registers, no VALU): 4 channels per stage into a double-buffered LDS tile (the same 33 KiB);
//   * lanes whose 16-byte piece lies inside the row: `global_load_lds_dwordx4`, 1 KiB per wave instruction, one per channel row;
//   * the columns no such lane covers (reflect padding at the row ends, the 8 tail columns): `global_load_lds_dword` with the reflected
//     source address per lane, wave w serving row w; columns that are zeros for every channel (beyond the padding) are zeroed once.
// Per stage: the stage's 48 LDS values into registers | DMA of stage s + 1 | 336 FMAs from registers under that DMA | vmcnt(0) + ONE barrier.
// (hipcc drains vmcnt before the first LDS access behind an LDS DMA: nothing touches LDS between the DMA issue and the end of the stage.)
// LDS column c' of a row = source column t0 - padL + c', as in the kernel above: a lane's window starts at its own 16-byte aligned column.
constexpr int C1P_CH = 4;

template <int K, int MO>
__global__ __launch_bounds__(256) void conv_fewout_rows_plain_kernel(const Cout1Args p) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    typedef const __attribute__((address_space(4))) float* cfp_t;
    __shared__ __attribute__((aligned(16))) float Xs[2][C1P_CH][C1L_ROW];
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, b = blockIdx.y;
    const int breal = p.Fo > 1 ? b / p.Fo : b, fo = p.Fo > 1 ? b - breal * p.Fo : 0;
    const int t0 = tile * C1L_TN, gbase = t0 - p.padL;               // source column of LDS column 0
    const float* s0 = p.src0 + (size_t)breal * p.in_sB0 + (size_t)fo * p.in_sB1;
    const int g0 = gbase + 4 * tid;
    const bool vec_ok = g0 >= 0 && g0 + 3 < p.T;                     // this lane's 16-byte piece lies inside the row
    // columns no 16-byte lane covers: [0, cl) and [cr, C1L_ROW)
    const int cl = gbase < 0 ? (((-gbase + 3) >> 2) << 2) : 0;
    int first_bad = (p.T - gbase) >> 2;                               // smallest tid with gbase + 4 tid + 3 >= T
    first_bad = first_bad < (cl >> 2) ? (cl >> 2) : (first_bad > 256 ? 256 : first_bad);
    const int cr = 4 * first_bad;
    const int refl = 2 * (p.Leff - 1);
    // this wave's edge columns (it serves row `wid` of every stage): column -> reflected source index, or -1 (a zero for every channel)
    constexpr int NE = (C1L_ROW + 63) / 64 + 1;                       // 64-column pieces: [0, cl) is one, [cr, C1L_ROW) at most 17
    int esrc[2];                                                      // common case: two pieces (left + right); longer ranges loop below
    auto edge_src = [&](int col) __attribute__((always_inline)) {
        const int g = gbase + col;
        int src = g < 0 ? -g : g;
        src = src >= p.Leff ? refl - src : src;
        return (col < C1L_ROW && g >= -p.padL && src >= 0 && src < p.T) ? src : -1;
    };
    (void)NE;
    esrc[0] = lane < cl ? edge_src(lane) : -1;
    esrc[1] = edge_src(cr + lane);
    // zeros: every edge column of both buffers once (the DMA below rewrites the ones that have a source, for every stage)
    for (int e = tid; e < 2 * C1P_CH * C1L_ROW; e += 256) {
        const int col = e % C1L_ROW;
        if (col < cl || col >= cr) (&Xs[0][0][0])[e] = 0.f;
    }
    __syncthreads();
    const int nstage = (p.Cin + C1P_CH - 1) / C1P_CH;
    auto stage_dma = [&](int st) __attribute__((always_inline)) {
        float* X = &Xs[st & 1][0][0];
        if (vec_ok) {
#pragma unroll
            for (int r = 0; r < C1P_CH; ++r) {
                const int c = st * C1P_CH + r < p.Cin ? st * C1P_CH + r : p.Cin - 1;      // past the last channel: a dropped re-read
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(s0 + (size_t)c * p.T + g0), (lds_ptr_t)(X + r * C1L_ROW + 256 * wid), 16, 0, 0);
            }
        }
        {   // edge columns of row `wid`
            const int c = st * C1P_CH + wid < p.Cin ? st * C1P_CH + wid : p.Cin - 1;
            const float* row = s0 + (size_t)c * p.T;
            float* Xr = X + wid * C1L_ROW;
            if (esrc[0] >= 0) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(row + esrc[0]), (lds_ptr_t)Xr, 4, 0, 0);
            if (esrc[1] >= 0) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(row + esrc[1]), (lds_ptr_t)(Xr + cr), 4, 0, 0);
            for (int base = cr + 64; base < C1L_ROW; base += 64) {    // rows much shorter than the tile
                const int src = edge_src(base + lane);
                if (src >= 0) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(row + src), (lds_ptr_t)(Xr + base), 4, 0, 0);
            }
        }
    };
    float acc[MO][4];
#pragma unroll
    for (int m = 0; m < MO; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
    stage_dma(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int st = 0; st < nstage; ++st) {
        f32x4 q[C1P_CH][3];
#pragma unroll
        for (int r = 0; r < C1P_CH; ++r)
#pragma unroll
            for (int v = 0; v < 3; ++v) q[r][v] = *(const f32x4*)&Xs[st & 1][r][4 * tid + 4 * v];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the values are in registers before the other buffer is rewritten
        if (st + 1 < nstage) stage_dma(st + 1);
#pragma unroll
        for (int r = 0; r < C1P_CH; ++r) {
            const int c = st * C1P_CH + r;
            if (c >= p.Cin) break;                                    // uniform
            float x[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) x[i] = q[r][i >> 2][i & 3];
            // the MO x K weights of the channel in one go (one wait per channel instead of one per output channel).  Uniform address,
            // CONSTANT address space: behind the asm barriers above hipcc no longer proves the weights unclobbered and would fetch them
            // with vector loads -- whose vmcnt waits then drain the DMA in front of the FMAs
            float wv[MO][K];
#pragma unroll
            for (int m = 0; m < MO; ++m) {
                const cfp_t wr = (cfp_t)(p.w + ((size_t)m * p.Cin + c) * K);
#pragma unroll
                for (int kk = 0; kk < K; ++kk) wv[m][kk] = wr[kk];
            }
#pragma unroll
            for (int m = 0; m < MO; ++m)
#pragma unroll
                for (int kk = 0; kk < K; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][j] = fmaf(wv[m][kk], x[j + kk], acc[m][j]);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // ---- epilogue: bias, store, statistics of the valid outputs (as above)
    float s1v = 0.f, s2v = 0.f;
#pragma unroll
    for (int m = 0; m < MO; ++m) {
        float o[4];
        const float bm = p.bias[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = acc[m][j] + bm;
            const int n = t0 + 4 * tid + j;
            if (n < p.T) { s1v += o[j]; s2v = fmaf(o[j], o[j], s2v); }
        }
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        float* orow = p.out + (size_t)breal * p.out_sB + (size_t)fo * p.out_sF + (size_t)m * p.out_sM + t0 + 4 * tid;
        if (t0 + 4 * tid + 3 < p.T) *(f32x4u*)orow = (f32x4){o[0], o[1], o[2], o[3]};
        else
#pragma unroll
            for (int j = 0; j < 4; ++j) if (t0 + 4 * tid + j < p.T) orow[j] = o[j];
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
            const size_t slot = ((size_t)breal * p.part_sB0 + (size_t)fo * gridDim.x + tile) * 2;
            p.partials[slot] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
            p.partials[slot + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        }
    }
}

constexpr int C1_WOUT = 248, C1_TN = 4 * C1_WOUT, C1_UN = 4;    // outputs per wave / per workgroup; channels per load group

// lane i <- lane i + 1 over the whole wavefront (DPP wave_shl:1; lane 63 gets 0)
__device__ __forceinline__ float wave_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// Streaming form: no LDS slab, no barrier in the channel loop.  A wave covers 248 consecutive outputs: every lane loads ONE 16-byte
// piece per source and channel (lanes 62, 63 only supply the k - 1 halo), applies the prologue to its 4 samples once, and gets the 6
// samples to its right from the next two lanes by whole-wave DPP shifts; the loads of the next 4 channels are in flight while the
// current 4 are multiplied.  Lanes at a padded edge (reflect, conv.py:82-99) gather their 4 samples by index instead.
template <int K, bool DUAL, int MO>
__global__ __launch_bounds__(256) void conv_cout1_kernel(const Cout1Args p) {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int breal = p.Fo > 1 ? b / p.Fo : b, fo = p.Fo > 1 ? b - breal * p.Fo : 0;
    const int t0 = tile * C1_TN + wid * C1_WOUT;               // first output of this wave
    const bool wave_on = t0 < p.T;                              // uniform per wave
    const size_t in_off = (size_t)breal * p.in_sB0 + (size_t)fo * p.in_sB1;
    const float* s0 = p.src0 + in_off;
    const float* s1 = DUAL ? p.src1 + in_off : s0;
    const bool wrap = p.affC != p.Cin;                          // virtual channel = frequency row * affC + real channel
    const float2* a0 = p.aff0 ? (const float2*)p.aff0 + (size_t)breal * p.affC : nullptr;
    const float2* a1 = (DUAL && p.aff1) ? (const float2*)p.aff1 + (size_t)breal * p.affC : nullptr;
    const int g0 = t0 - p.padL + 4 * lane;                      // padded-coordinate index of this lane's first sample
    const bool vec_ok = g0 >= 0 && g0 + 3 < p.T;
    int esrc[4]; unsigned emask = 0;
    {
        const int refl = 2 * (p.Leff - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = g0 + j;
            int src = g < 0 ? -g : g;
            src = src >= p.Leff ? refl - src : src;
            const bool ok = g >= -p.padL && src >= 0 && src < p.T;
            esrc[j] = ok ? src : 0;
            emask |= (ok ? 1u : 0u) << j;
        }
    }
    const unsigned vmask = vec_ok ? 0xFu : emask;
    float acc[MO][4];
#pragma unroll
    for (int m = 0; m < MO; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;

    // straight-line loads (no branch, clamped addresses): a conditional load makes hipcc wait for it right behind its issue.  Lanes at a
    // padded edge re-gather their samples in mul_group (only the first / last wave of a row has such lanes)
    const int gsafe = g0 < 0 ? 0 : (g0 + 3 < p.T ? g0 : (p.T >= 4 ? p.T - 4 : 0));
    auto load_group = [&](int c0, f32x4 (&r0)[C1_UN], f32x4 (&r1)[DUAL ? C1_UN : 1]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < C1_UN; ++u) {
            const int c = c0 + u < p.Cin ? c0 + u : p.Cin - 1;
            r0[u] = *(const f32x4u*)(s0 + (size_t)c * p.T + gsafe);
            if (DUAL) r1[u] = *(const f32x4u*)(s1 + (size_t)c * p.T + gsafe);
        }
    };
    auto mul_group = [&](int c0, const f32x4 (&r0)[C1_UN], const f32x4 (&r1)[DUAL ? C1_UN : 1]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < C1_UN; ++u) {
            const int c = c0 + u;
            if (c >= p.Cin) break;                                                // uniform
            const int ca = wrap ? c % p.affC : c;
            const float2 A = a0 ? a0[ca] : make_float2(1.f, 0.f);
            const float2 A1 = a1 ? a1[ca] : make_float2(1.f, 0.f);
            f32x4 w0 = r0[u], w1 = DUAL ? r1[u] : r0[u];
            if (!vec_ok) {                                                        // padded edge: gather by index
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w0[j] = s0[(size_t)c * p.T + esrc[j]];
                    if (DUAL) w1[j] = s1[(size_t)c * p.T + esrc[j]];
                }
            }
            float x[12];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = w0[j];
                if (!FC_ABL(p.ablate, 8)) {
                    v = fmaf(v, A.x, A.y);
                    if (DUAL) v = v + fmaf(w1[j], A1.x, A1.y);
                    if (p.elu) v = elu_f(v, p.alpha);
                }
                x[j] = ((vmask >> j) & 1u) ? v : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) x[4 + j] = wave_shl1(x[j]);               // samples of lane + 1
#pragma unroll
            for (int j = 0; j < 2; ++j) x[8 + j] = wave_shl1(x[4 + j]);           // first two samples of lane + 2
            if FC_ABL(p.ablate, 1) continue;
#pragma unroll
            for (int m = 0; m < MO; ++m) {
                const float* wr = p.w + ((size_t)m * p.Cin + c) * K;             // uniform address: scalar loads
#pragma unroll
                for (int kk = 0; kk < K; ++kk) {
                    const float wv = wr[kk];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][j] = fmaf(wv, x[j + kk], acc[m][j]);
                }
            }
        }
    };
    if (wave_on) {
        f32x4 ra0[C1_UN], ra1[DUAL ? C1_UN : 1], rb0[C1_UN], rb1[DUAL ? C1_UN : 1];
        load_group(0, ra0, ra1);
        for (int c0 = 0; c0 < p.Cin; c0 += 2 * C1_UN) {       // the loads past the last channel re-read it (clamped) and are dropped
            load_group(c0 + C1_UN, rb0, rb1);
            mul_group(c0, ra0, ra1);
            load_group(c0 + 2 * C1_UN, ra0, ra1);
            mul_group(c0 + C1_UN, rb0, rb1);
        }
    }
    // ---- epilogue: bias, store, statistics of the valid outputs (lanes 62, 63 hold no outputs)
    float s1v = 0.f, s2v = 0.f;
    const int n0 = t0 + 4 * lane;
    const bool out_lane = wave_on && lane < C1_WOUT / 4;
#pragma unroll
    for (int m = 0; m < MO; ++m) {
        float o[4];
        const float bm = p.bias[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = acc[m][j] + bm;
            if (out_lane && n0 + j < p.T) { s1v += o[j]; s2v = fmaf(o[j], o[j], s2v); }
        }
        if (FC_ABL(p.ablate, 2) || !out_lane) continue;
        float* orow = p.out + (size_t)breal * p.out_sB + (size_t)fo * p.out_sF + (size_t)m * p.out_sM + n0;
        if (n0 + 3 < p.T) *(f32x4u*)orow = (f32x4){o[0], o[1], o[2], o[3]};
        else
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n0 + j < p.T) orow[j] = o[j];
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
            const size_t slot = ((size_t)breal * p.part_sB0 + (size_t)fo * gridDim.x + tile) * 2;
            p.partials[slot] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
            p.partials[slot + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        }
    }
}

bool conv_cout1_ok(const ConvLaunch& c) {
    return c.w_plain && c.M >= 1 && c.M <= 4 && c.store_lo <= 0 && c.store_hi >= c.Fo && c.stride == 1 && c.dil == 1 && !c.up_r && !c.pad_zero && !c.s0.div && c.out_sT == 1 && c.Tout == c.Tin &&
           (c.k == 7 || c.k == 3 || c.k == 5) && c.padL + c.padR == c.k - 1;
}

static hipError_t launch_conv_cout1(const ConvLaunch& c, hipStream_t st) {
    Cout1Args a;
    a.src0 = c.s0.ptr; a.aff0 = c.s0.aff; a.src1 = c.s1.ptr; a.aff1 = c.s1.aff;
    a.w = c.w_plain; a.bias = c.bias; a.out = c.out; a.partials = c.partials;
    a.Cin = c.Cin; a.T = c.Tin; a.k = c.k; a.padL = c.padL;
    a.Fo = c.Fo > 1 ? c.Fo : 1; a.affC = c.affC > 0 ? c.affC : c.Cin;
    a.in_sB0 = c.in_sB0 ? c.in_sB0 : (long long)c.Cin * c.Tin; a.in_sB1 = c.in_sB1;
    a.out_sB = c.out_sB; a.out_sF = c.out_sF; a.out_sM = c.out_sM;
    const int tile_n = conv_fewout_tile(c);
    a.part_sB0 = c.part_sB0 ? c.part_sB0 : (long long)a.Fo * ceil_div(c.Tout, tile_n);
    static const int ablate_env = ab_knob("FC_ABLATE", 0);
    a.ablate = ablate_env;
    const int maxpad = c.padL > c.padR ? c.padL : c.padR;
    a.Leff = c.Tin > maxpad ? c.Tin : maxpad + 1;
    a.elu = c.elu; a.alpha = c.alpha;
    if (c.B > 65535) return hipErrorInvalidValue;
    dim3 grid(ceil_div(c.Tout, tile_n), c.B), block(256);
    const bool rows = tile_n == C1L_TN;
    // one materialised source, no prologue arithmetic: the all-DMA staging form
    if (conv_fewout_plain(c)) {
#define FC_C1P(KK, MM) if (c.k == KK && c.M == MM) { hipLaunchKernelGGL((conv_fewout_rows_plain_kernel<KK, MM>), grid, block, 0, st, a); return hipGetLastError(); }
        FC_C1P(3, 1) FC_C1P(3, 2) FC_C1P(3, 3) FC_C1P(3, 4) FC_C1P(5, 1) FC_C1P(5, 2) FC_C1P(5, 3) FC_C1P(5, 4)
        FC_C1P(7, 1) FC_C1P(7, 2) FC_C1P(7, 3) FC_C1P(7, 4)
#undef FC_C1P
    }
#define FC_C1M(KK, MM)                                                                                  \
    case MM:                                                                                            \
        if (rows) {                                                                                     \
            if (c.s1.ptr) hipLaunchKernelGGL((conv_fewout_rows_kernel<KK, true, MM>), grid, block, 0, st, a);  \
            else hipLaunchKernelGGL((conv_fewout_rows_kernel<KK, false, MM>), grid, block, 0, st, a);   \
        } else {                                                                                        \
            if (c.s1.ptr) hipLaunchKernelGGL((conv_cout1_kernel<KK, true, MM>), grid, block, 0, st, a); \
            else hipLaunchKernelGGL((conv_cout1_kernel<KK, false, MM>), grid, block, 0, st, a);         \
        }                                                                                               \
        break;
#define FC_C1(KK)                                                                                       \
    case KK:                                                                                            \
        switch (c.M) { FC_C1M(KK, 1) FC_C1M(KK, 2) FC_C1M(KK, 3) FC_C1M(KK, 4) default: return hipErrorInvalidValue; } \
        break;
    switch (c.k) {
        FC_C1(3)
        FC_C1(5)
        FC_C1(7)
        default: return hipErrorInvalidValue;
    }
#undef FC_C1
#undef FC_C1M
    return hipGetLastError();
}

// =================================================================================================
// 1c. Fused head of a THIN residual block (C = 32 or 64 channels, the HBM-bound end of the nets):
//         sc = shortcut(x)           Conv1d(C -> C, k = 1)            on x
//         b1 = block.1(ELU(x))       Conv1d(C -> C/2, k = K3, dil)    on ELU(x), reflect padded
//     (SEANetResnetBlock.forward seanet_encoder.py:44-61; x = the block's input with its pending GroupNorm affine(s) applied,
//     one tensor or the sum of two).  Both convs read the same activation: as two launches of the general kernel the input
//     (the largest tensors of the whole path) is streamed from HBM twice, and the 16-row block.1 is padded to a 32-row MFMA tile.
//     Here one workgroup stages a [32 channels][128 + halo columns] slab ONCE into LDS in two views (affine'd, and ELU'd),
//     both weight matrices stay resident in LDS for the workgroup's whole tile range, and each of the 4 waves computes 32
//     columns of BOTH outputs: shortcut on 32x32x2 MFMAs, block.1 on 16x16x4 MFMAs when it has 16 rows (no padded rows).
//     Next item's global loads (16-byte, one channel row per 32 lanes) are issued right after the registers were consumed and
//     fly during the MFMA phase; 2-3 workgroups per CU cover each other's barriers.  Same epilogue contract as the general
//     kernel: raw outputs + one deterministic fp64 (sum, sum of squares) partial per (utterance, tile) and output.
//     Accumulation order per output element: bias, then channels ascending (shortcut) / (tap, channel) ascending (block.1).
// =================================================================================================
struct ResHeadArgs {
    const float *src0, *aff0, *src1, *aff1;    // [B][C][T], per-(b,c) affine or null
    const float *wsc, *wb1;                    // LDS images: wsc[c][m] (C x C), wb1[kk*C + c][h] (K3*C x C/2)
    const float *bsc, *bb1;                    // biases [C], [C/2]
    float *out_sc, *out_b1;                    // [B][C][T], [B][C/2][T]
    double *part_sc, *part_b1;                 // [B][ntiles][2] or null
    int T, padL, padR, dil, Leff, ntiles;
    float alpha;
};
constexpr int RH_BN = 128, RH_XS = 144, RH_CH = 32;      // columns per tile, LDS row stride (== 16 mod 32), channels per chunk

// sum over the 64 lanes of a wave in a FIXED order: 16-lane rows by DPP butterflies (quad swaps, half-row and row mirrors), then
// the four row totals added in row order.  ~20 VALU instructions (a 64-bit __shfl_xor tree is 12 ds_bpermute + 6 v_add_f64).
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return ((r0 + r1) + r2) + r3;
}

template <int C, int K3, bool DUAL>
__global__ __launch_bounds__(256, C == 32 ? 3 : 2) void reshead_kernel(const ResHeadArgs p) {
    constexpr int HID = C / 2, NCH = C / RH_CH, MT = C / 32;
    constexpr bool B16 = HID == 16;                         // block.1 on 16x16x4 tiles
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wsc = smem;                                      // [C][C]
    float* Wb1 = Wsc + C * C;                               // [K3*C][HID]
    float* Xa = Wb1 + K3 * C * HID;                         // [32][XS]  affine'd input (shortcut operand)
    float* Xe = Xa + RH_CH * RH_XS;                         // [32][XS]  ELU'd input (block.1 operand)
    float* bias_s = Xe + RH_CH * RH_XS;                     // [C + HID]
    float* red = bias_s + C + HID;                          // [2 outputs][4 waves][2]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31, g4 = lane >> 4, r16 = lane & 15;
    const int b = blockIdx.y;
    const int t_begin = (int)(((long long)p.ntiles * blockIdx.x) / gridDim.x);
    const int t_end = (int)(((long long)p.ntiles * (blockIdx.x + 1)) / gridDim.x);
    if (t_begin >= t_end) return;

    // ---- resident weights and biases
    for (int i = tid * 4; i < C * C; i += 1024) *(f32x4*)(Wsc + i) = *(const f32x4*)(p.wsc + i);
    for (int i = tid * 4; i < K3 * C * HID; i += 1024) *(f32x4*)(Wb1 + i) = *(const f32x4*)(p.wb1 + i);
    for (int i = tid; i < C + HID; i += 256) bias_s[i] = i < C ? p.bsc[i] : p.bb1[i - C];

    const int halo = (K3 - 1) * p.dil;                      // <= 8 (host)
    const int OFF = (4 - (p.padL & 3)) & 3;                 // main columns start 16-byte aligned in LDS
    // my elements of a chunk: 4 rows (8 apart) x 4 consecutive main columns, + one halo element for the first 32*halo threads
    const int c4 = tid & 31, row0 = tid >> 5;
    const bool has_h = tid < RH_CH * halo;
    const int h_row = has_h ? tid / halo : 0, h_j = has_h ? tid - h_row * halo : 0;
    const int h_col = has_h ? (h_j < p.padL ? h_j - p.padL : RH_BN + (h_j - p.padL)) : 0;   // slab column relative to n0
    const size_t ubase = (size_t)b * C * p.T;
    const float* s0 = p.src0 + ubase;
    const float* s1 = DUAL ? p.src1 + ubase : s0;
    // the producers' GroupNorm affine of MY rows, once per workgroup (a global load per row inside the item loop costs an L2
    // round trip on the critical path and, worse, a vmcnt(0) that drains every prefetch and store in flight)
    float2 A0[NCH][4], A1[DUAL ? NCH : 1][DUAL ? 4 : 1], Ah0[NCH], Ah1[DUAL ? NCH : 1];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = ch * RH_CH + row0 + 8 * r;
            A0[ch][r] = p.aff0 ? ((const float2*)p.aff0)[(size_t)b * C + c] : make_float2(1.f, 0.f);
            if (DUAL) A1[ch][r] = p.aff1 ? ((const float2*)p.aff1)[(size_t)b * C + c] : make_float2(1.f, 0.f);
        }
        const int c = ch * RH_CH + h_row;
        Ah0[ch] = p.aff0 ? ((const float2*)p.aff0)[(size_t)b * C + c] : make_float2(1.f, 0.f);
        if (DUAL) Ah1[ch] = p.aff1 ? ((const float2*)p.aff1)[(size_t)b * C + c] : make_float2(1.f, 0.f);
    }
    const int refl = 2 * (p.Leff - 1);
    auto resolve = [&](int g, bool& ok) __attribute__((always_inline)) {      // global column -> source index (pad1d reflect, conv.py:82-99)
        ok = g >= -p.padL && g < p.T + p.padR;
        int src = g < 0 ? -g : g;
        src = src >= p.Leff ? refl - src : src;
        ok = ok && src >= 0 && src < p.T;
        return ok ? src : 0;
    };

    // TWO register sets: the loads of item f+2 are issued while those of item f+1 are still in flight, so every load has two
    // item times to come back and a workgroup keeps 2 x 16 KiB (x2 with two sources) of reads outstanding.
    f32x4 v0[2][4], v1[2][DUAL ? 4 : 1];
    float hv0[2] = {0.f, 0.f}, hv1[2] = {0.f, 0.f};
    unsigned vmask[2] = {0u, 0u};                            // GENERIC path: validity of my 16 main elements + bit 16 the halo element
    int ld_tile = 0, ld_chunk = 0, st_chunk = 0;
    // FAST = every tile of the segment is a full interior tile: no padding, no masks, no per-lane conditions around memory
    // instructions -> the whole item is straight-line code and the compiler's vmcnt waits are exact (conditional loads / stores
    // made it fall back to vmcnt(0) in front of every use: each wait then drained the prefetch AND the previous tile's stores)
    auto load_item = [&](auto set_tag, auto fast_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;
        const int n0 = ld_tile * RH_BN, c0 = ld_chunk * RH_CH;
        if (++ld_chunk == NCH) { ld_chunk = 0; ++ld_tile; }
        if (FAST) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = (size_t)(c0 + row0 + 8 * r) * p.T + n0 + 4 * c4;
                v0[S][r] = *(const f32x4u*)(s0 + o);
                if (DUAL) v1[S][r] = *(const f32x4u*)(s1 + o);
            }
            const size_t o = (size_t)(c0 + h_row) * p.T + n0 + h_col;     // lanes without a halo element re-read column n0 of row c0
            hv0[S] = s0[o];
            if (DUAL) hv1[S] = s1[o];
        } else {
            unsigned vm = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t ro = (size_t)(c0 + row0 + 8 * r) * p.T;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bool ok;
                    const int src = resolve(n0 + 4 * c4 + j, ok);
                    v0[S][r][j] = s0[ro + src];
                    if (DUAL) v1[S][r][j] = s1[ro + src];
                    vm |= (ok ? 1u : 0u) << (4 * r + j);
                }
            }
            bool ok;
            const int src = resolve(n0 + h_col, ok);
            const size_t o = (size_t)(c0 + h_row) * p.T + src;
            hv0[S] = s0[o];
            if (DUAL) hv1[S] = s1[o];
            vm |= (ok ? 1u : 0u) << 16;
            vmask[S] = vm;
        }
    };
    auto stage_item = [&](auto set_tag, auto fast_tag) __attribute__((always_inline)) {   // registers -> (affine, + second source) -> Xa ; ELU -> Xe
        constexpr int S = decltype(set_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;
        const int ch = st_chunk;
        if (++st_chunk == NCH) st_chunk = 0;
        const unsigned vm = FAST ? 0x1ffffu : vmask[S];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 A = NCH == 1 ? A0[0][r] : (ch ? A0[NCH - 1][r] : A0[0][r]);
            const float2 B1 = !DUAL ? make_float2(1.f, 0.f) : (NCH == 1 ? A1[0][DUAL ? r : 0] : (ch ? A1[DUAL ? NCH - 1 : 0][DUAL ? r : 0] : A1[0][DUAL ? r : 0]));
            f32x4 xa, xe;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = fmaf(v0[S][r][j], A.x, A.y);
                if (DUAL) v = v + fmaf(v1[S][DUAL ? r : 0][j], B1.x, B1.y);
                float e = elu_f(v, p.alpha);
                if (!FAST) { const bool ok = (vm >> (4 * r + j)) & 1u; v = ok ? v : 0.f; e = ok ? e : 0.f; }
                xa[j] = v; xe[j] = e;
            }
            const int o = (row0 + 8 * r) * RH_XS + OFF + p.padL + 4 * c4;
            *(f32x4*)(Xa + o) = xa;
            *(f32x4*)(Xe + o) = xe;
        }
        {
            const float2 A = NCH == 1 ? Ah0[0] : (ch ? Ah0[NCH - 1] : Ah0[0]);
            const float2 B1 = !DUAL ? make_float2(1.f, 0.f) : (NCH == 1 ? Ah1[0] : (ch ? Ah1[DUAL ? NCH - 1 : 0] : Ah1[0]));
            float v = fmaf(hv0[S], A.x, A.y);
            if (DUAL) v = v + fmaf(hv1[S], B1.x, B1.y);
            float e = elu_f(v, p.alpha);
            if (!FAST) e = ((vm >> 16) & 1u) ? e : 0.f;
            if (has_h) Xe[h_row * RH_XS + OFF + p.padL + h_col] = e;        // the shortcut (k = 1) never reads halo columns
        }
    };

    // accumulators: shortcut MT x (32 rows x 32 cols); block.1 one 32x32 tile or two 16x16 tiles
    f32x16 asc[MT];
    f32x16 ab32;
    f32x4 ab16[2];
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) asc[mt][r] = bias_s[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];
        if (B16) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) ab16[j][r] = bias_s[C + 4 * g4 + r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ab32[r] = bias_s[C + (r & 3) + 8 * (r >> 2) + 4 * hi];
        }
    };
    const int nl0 = wid * 32;
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    __syncthreads();                                        // weights / biases visible
    init_acc();

    // one item = one 32-channel chunk of one 128-column tile
    auto item = [&](int f, int nitems, int& tile, int& chunk, auto set_tag, auto fast_tag) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        stage_item(set_tag, fast_tag);
        if (f + 2 < nitems) load_item(set_tag, fast_tag);   // refill the set just consumed: in flight for two item times
        __syncthreads();
        const int c0 = chunk * RH_CH;
        {   // ---- shortcut: sc += Wsc[:, c0 .. c0+31] . Xa
            const float* xb = Xa + hi * RH_XS + OFF + p.padL + nl0 + l31;
            const float* wa = Wsc + (c0 + hi) * C + l31;
#pragma unroll 2
            for (int ks = 0; ks < 16; ++ks) {
                const float bv = xb[2 * ks * RH_XS];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    asc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * ks * C + mt * 32], bv, asc[mt], 0, 0, 0);
            }
        }
        if (B16) {   // ---- block.1, 16 rows: 16x16x4 tiles, 4 channels per k-step, two 16-column groups
#pragma unroll 1
            for (int kk = 0; kk < K3; ++kk) {
                const float* xb = Xe + g4 * RH_XS + OFF + nl0 + r16 + kk * p.dil;
                const float* wa = Wb1 + (kk * C + c0 + g4) * HID + r16;
#pragma unroll 4
                for (int q = 0; q < 8; ++q) {
                    const float av = wa[4 * q * HID];
                    ab16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xb[4 * q * RH_XS], ab16[0], 0, 0, 0);
                    ab16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xb[4 * q * RH_XS + 16], ab16[1], 0, 0, 0);
                }
            }
        } else {     // ---- block.1, 32 rows
#pragma unroll 1
            for (int kk = 0; kk < K3; ++kk) {
                const float* xb = Xe + hi * RH_XS + OFF + nl0 + l31 + kk * p.dil;
                const float* wa = Wb1 + (kk * C + c0 + hi) * HID + l31;
#pragma unroll 4
                for (int ks = 0; ks < 16; ++ks)
                    ab32 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * ks * HID], xb[2 * ks * RH_XS], ab32, 0, 0, 0);
            }
        }
        const bool tile_done = chunk == NCH - 1;
        if (tile_done) {
            // ---- epilogue: raw stores + per-lane statistics of the valid columns
            const int n0 = tile * RH_BN;
            float s1 = 0.f, s2 = 0.f, u1 = 0.f, u2 = 0.f;
            {
                const int col = n0 + nl0 + l31;
                const bool okc = FAST || col < p.T;
                // one lane pointer (column + the lane's row offset); every accumulator row is a wave-uniform offset from it
                float* o = p.out_sc + ubase + col + (size_t)(4 * hi) * p.T;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const size_t mo = (size_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * p.T;
                        const float v = asc[mt][r];
                        if (FAST) { o[mo] = v; s1 += v; s2 = fmaf(v, v, s2); }
                        else if (okc) { o[mo] = v; s1 += v; s2 = fmaf(v, v, s2); }
                    }
            }
            float* ob = p.out_b1 + (size_t)b * HID * p.T;
            if (B16) {
                const int col = n0 + nl0 + r16;
                float* o = ob + col + (size_t)(4 * g4) * p.T;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = ab16[j][r];
                        const size_t mo = (size_t)r * p.T + 16 * j;
                        if (FAST) { o[mo] = v; u1 += v; u2 = fmaf(v, v, u2); }
                        else if (col + 16 * j < p.T) { o[mo] = v; u1 += v; u2 = fmaf(v, v, u2); }
                    }
                }
            } else {
                const int col = n0 + nl0 + l31;
                float* o = ob + col + (size_t)(4 * hi) * p.T;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = ab32[r];
                    const size_t mo = (size_t)((r & 3) + 8 * (r >> 2)) * p.T;
                    if (FAST) { o[mo] = v; u1 += v; u2 = fmaf(v, v, u2); }
                    else if (col < p.T) { o[mo] = v; u1 += v; u2 = fmaf(v, v, u2); }
                }
            }
            if (p.part_sc) {                                // wave totals in a fixed order (fp32: <= 24 values per lane, 64 lanes)
                const float w0 = wave_sum_f32(s1), w1 = wave_sum_f32(s2), w2 = wave_sum_f32(u1), w3 = wave_sum_f32(u2);
                if (lane == 0) *(f32x4*)(red + 4 * wid) = (f32x4){w0, w1, w2, w3};
            }
            init_acc();
        }
        __syncthreads();                                    // Xa / Xe free for the next item; red visible
        if (tile_done) {
            if (p.part_sc && tid < 4) {                     // fp64 sum of the 4 wave totals in wave order: one partial per (utterance, tile)
                const int outp = tid >> 1, comp = tid & 1;
                double* dst = outp ? p.part_b1 : p.part_sc;
                const float* rr = red + outp * 2 + comp;
                dst[((size_t)b * p.ntiles + tile) * 2 + comp] = (((double)rr[0] + (double)rr[4]) + (double)rr[8]) + (double)rr[12];
            }
            ++tile; chunk = 0;
        } else {
            ++chunk;
        }
    };
    // a segment = consecutive tiles handled by one code path; its pipeline is self-contained (prologue loads of its first two
    // items, nothing prefetched past its end)
    auto run = [&](int ta, int tb, auto fast_tag) __attribute__((always_inline)) {
        if (ta >= tb) return;
        const int nitems = (tb - ta) * NCH;
        ld_tile = ta; ld_chunk = 0; st_chunk = 0;
        int tile = ta, chunk = 0;
        load_item(Set0(), fast_tag);
        if (nitems > 1) load_item(Set1(), fast_tag);
        for (int f = 0; f < nitems; f += 2) {
            item(f, nitems, tile, chunk, Set0(), fast_tag);
            if (f + 1 < nitems) item(f + 1, nitems, tile, chunk, Set1(), fast_tag);
        }
    };
    // interior tiles: slab [n0 - padL, n0 + BN + padR) inside [0, T)
    int e0 = (p.padL + RH_BN - 1) / RH_BN;                                       // first tile with n0 - padL >= 0
    int e1 = p.T - p.padR - RH_BN >= 0 ? (p.T - p.padR - RH_BN) / RH_BN + 1 : 0;  // one past the last tile with n0 + BN + padR <= T
    if (e1 < e0) e1 = e0;
    const int f0 = e0 < t_begin ? t_begin : (e0 > t_end ? t_end : e0);
    const int f1 = e1 < f0 ? f0 : (e1 > t_end ? t_end : e1);
    run(t_begin, f0, std::false_type());
    run(f0, f1, std::true_type());
    run(f1, t_end, std::false_type());
}

size_t reshead_lds_bytes(int C, int K3) {
    const int HID = C / 2;
    return (size_t)(C * C + K3 * C * HID + 2 * RH_CH * RH_XS + C + HID + 16) * sizeof(float);
}
int reshead_ntiles(int T) { return ceil_div(T, RH_BN); }
bool reshead_ok(int C, int hid, int k_sc, int k_b1, int dil, int stride) {
    return (C == 32 || C == 64) && hid * 2 == C && k_sc == 1 && stride == 1 && (k_b1 == 3 || k_b1 == 5 || k_b1 == 7) && (k_b1 - 1) * dil <= 8;
}

template <int C, int K3, bool DUAL>
static hipError_t launch_reshead_t(const ResHeadArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0ull};
    auto kfn = reshead_kernel<C, K3, DUAL>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        hipError_t ea = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return ea;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_reshead(const ResHeadLaunch& c, hipStream_t st) {
    if (!reshead_ok(c.C, c.C / 2, 1, c.k, c.dil, 1)) return hipErrorInvalidValue;
    ResHeadArgs a;
    a.src0 = c.s0.ptr; a.aff0 = c.s0.aff; a.src1 = c.s1.ptr; a.aff1 = c.s1.aff;
    a.wsc = c.wsc; a.wb1 = c.wb1; a.bsc = c.bsc; a.bb1 = c.bb1;
    a.out_sc = c.out_sc; a.out_b1 = c.out_b1; a.part_sc = c.part_sc; a.part_b1 = c.part_b1;
    a.T = c.T; a.padL = c.padL; a.padR = c.padR; a.dil = c.dil;
    const int maxpad = c.padL > c.padR ? c.padL : c.padR;
    a.Leff = c.T > maxpad ? c.T : maxpad + 1;
    a.ntiles = reshead_ntiles(c.T);
    a.alpha = c.alpha;
    const size_t lds = reshead_lds_bytes(c.C, c.k);
    const int per_cu = (int)((160 * 1024) / lds) > 3 ? 3 : (int)((160 * 1024) / lds);
    static const int target_env = ab_knob("FC_RH_WGS", 0);
    const int target = target_env ? target_env : 256 * (per_cu < 1 ? 1 : per_cu);
    int G = target / c.B;
    if (G < 1) G = 1;
    if (G > a.ntiles) G = a.ntiles;
    dim3 grid(G, c.B);
    const bool dual = c.s1.ptr != nullptr;
#define FC_RH(CC, KK)                                                                              \
    if (c.C == CC && c.k == KK)                                                                    \
        return dual ? launch_reshead_t<CC, KK, true>(a, grid, lds, st) : launch_reshead_t<CC, KK, false>(a, grid, lds, st);
    FC_RH(32, 3) FC_RH(64, 3) FC_RH(32, 5) FC_RH(64, 5) FC_RH(32, 7) FC_RH(64, 7)
#undef FC_RH
    return hipErrorInvalidValue;
}

// =================================================================================================
// 2. GroupNorm(1, C) statistics finalisation (nn.GroupNorm(num_groups=1), conv.py:45-52)
//    Fixed-order fp64 reduction of the per-workgroup partials -> mean / rstd -> per-(b,c) affine table
//    consumed by the next layer's prologue:  y = x*a + s,  a = rstd*gamma_c,  s = beta_c - mean*a.
// =================================================================================================
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ partials, int nblk, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int C, float eps, float* __restrict__ aff) {
    __shared__ double sh[2][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < nblk; i += 256) {
        s1 += partials[((size_t)b * nblk + i) * 2];
        s2 += partials[((size_t)b * nblk + i) * 2 + 1];
    }
    sh[0][tid] = s1; sh[1][tid] = s2;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
        __syncthreads();
    }
    const double mean = sh[0][0] / count;
    double var = sh[1][0] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    for (int c = tid; c < C; c += 256) {
        const float a = rstd * gamma[c];
        const float s = fmaf(-a, meanf, beta[c]);
        ((float2*)aff)[(size_t)b * C + c] = make_float2(a, s);
    }
}

hipError_t launch_gn_finalize(const double* partials, int nblk, double count, const float* gamma, const float* beta,
                              int C, float eps, int B, float* aff, hipStream_t st) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, st, partials, nblk, count, gamma, beta, C, eps, aff);
    return hipGetLastError();
}

// =================================================================================================
// 3. Elementwise materialisation (only where a tensor leaves the engine or changes layout)
// =================================================================================================
__global__ __launch_bounds__(256) void combine_kernel(Src s0, Src s1, int elu, float alpha, const float* __restrict__ mul,
                                                      int C, int Tsrc, int Tcopy, float* __restrict__ out,
                                                      long long o_sB, long long o_sC, long long o_sT) {
    const int b = blockIdx.z, c = blockIdx.y;
    const size_t row = (size_t)b * C + c;
    float2 a0 = make_float2(1.f, 0.f), a1 = make_float2(1.f, 0.f);
    if (s0.aff) a0 = ((const float2*)s0.aff)[row];
    if (s1.ptr && s1.aff) a1 = ((const float2*)s1.aff)[row];
    const float m = mul ? mul[b] : 1.f;
    if (o_sT == 1 && !s0.div) {
        // contiguous output rows (every materialisation inside the engine): 4 samples per thread, 16-byte loads / stores
        // (dword alignment only: rows start at row * Tsrc floats)
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        float* orow = out + (size_t)b * o_sB + (size_t)c * o_sC;
        const float* r0 = s0.ptr + row * Tsrc;
        const float* r1 = s1.ptr ? s1.ptr + row * Tsrc : r0;
        const int T4 = Tcopy >> 2;
        for (int q = blockIdx.x * 256 + threadIdx.x; q < T4; q += gridDim.x * 256) {
            const f32x4 x0 = *(const f32x4u*)(r0 + 4 * q);
            const f32x4 x1 = s1.ptr ? (f32x4)(*(const f32x4u*)(r1 + 4 * q)) : x0;
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = x0[j];
                if (s0.aff) v = fmaf(v, a0.x, a0.y);
                if (s1.ptr) {
                    float w = x1[j];
                    if (s1.aff) w = fmaf(w, a1.x, a1.y);
                    v = v + w;
                }
                if (elu) v = elu_f(v, alpha);
                if (mul) v = v * m;
                y[j] = v;
            }
            *(f32x4u*)(orow + 4 * q) = y;
        }
        for (int t = 4 * T4 + blockIdx.x * 256 + threadIdx.x; t < Tcopy; t += gridDim.x * 256) {
            float v = r0[t];
            if (s0.aff) v = fmaf(v, a0.x, a0.y);
            if (s1.ptr) {
                float w = r1[t];
                if (s1.aff) w = fmaf(w, a1.x, a1.y);
                v = v + w;
            }
            if (elu) v = elu_f(v, alpha);
            if (mul) v = v * m;
            orow[t] = v;
        }
        return;
    }
    for (int t = blockIdx.x * 256 + threadIdx.x; t < Tcopy; t += gridDim.x * 256) {
        float v = s0.ptr[row * Tsrc + t];
        if (s0.div) v = v / s0.div[b];
        if (s0.aff) v = fmaf(v, a0.x, a0.y);
        if (s1.ptr) {
            float w = s1.ptr[row * Tsrc + t];
            if (s1.aff) w = fmaf(w, a1.x, a1.y);
            v = v + w;
        }
        if (elu) v = elu_f(v, alpha);
        if (mul) v = v * m;
        out[(size_t)b * o_sB + (size_t)c * o_sC + (size_t)t * o_sT] = v;
    }
}

hipError_t launch_combine(const Src& s0, const Src& s1, int elu, float alpha, const float* mul, int B, int C, int Tsrc,
                          int Tcopy, float* out, long long o_sB, long long o_sC, long long o_sT, hipStream_t st) {
    if (Tcopy <= 0 || B <= 0) return hipSuccess;
    int gx = ceil_div(Tcopy, 256 * 4);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(combine_kernel, dim3(gx, C, B), dim3(256), 0, st, s0, s1, elu, alpha, mul, C, Tsrc, Tcopy, out,
                       o_sB, o_sC, o_sT);
    return hipGetLastError();
}

// Materialisation for the DMA-staged conv form (conv_kernel.h MODE 5): 4 channels interleaved, padding in place.  One thread = 4 channels x 4
// padded columns: four 16-byte loads (interior) or 16 resolved dword loads (the few columns around the row ends), four 16-byte stores = 64
// contiguous bytes.
__global__ __launch_bounds__(256) void combine_xq_kernel(Src s0, Src s1, int elu, float alpha, int C, int Tin, int padL, int padR, int pad_zero,
                                                         int Leff, float* __restrict__ xq) {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const int b = blockIdx.z, g4 = blockIdx.y, Tp = padL + Tin + padR;
    const size_t row0 = (size_t)b * C + 4 * g4;
    float2 a0[4], a1[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        a0[s4] = s0.aff ? ((const float2*)s0.aff)[row0 + s4] : make_float2(1.f, 0.f);
        a1[s4] = (s1.ptr && s1.aff) ? ((const float2*)s1.aff)[row0 + s4] : make_float2(1.f, 0.f);
    }
    const float* r0 = s0.ptr + row0 * Tin;
    const float* r1 = s1.ptr ? s1.ptr + row0 * Tin : r0;
    f32x4* orow = (f32x4*)xq + ((size_t)b * (C >> 2) + g4) * (size_t)Tp;
    const int hi_lim = Tin + padR, refl = 2 * (Leff - 1);
    auto act = [&](float v, float w, int s4) __attribute__((always_inline)) {
        if (s0.aff) v = fmaf(v, a0[s4].x, a0[s4].y);
        if (s1.ptr) v = v + (s1.aff ? fmaf(w, a1[s4].x, a1[s4].y) : w);
        if (elu) v = elu_f(v, alpha);
        return v;
    };
    const int T4 = (Tp + 3) >> 2;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < T4; q += gridDim.x * 256) {
        const int g = 4 * q - padL;                  // tensor column of padded column 4 q
        f32x4 x0[4], x1[4];
        bool ok[4] = {true, true, true, true};
        if (g >= 0 && g + 3 < Tin) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                x0[s4] = *(const f32x4u*)(r0 + (size_t)s4 * Tin + g);
                x1[s4] = s1.ptr ? (f32x4)(*(const f32x4u*)(r1 + (size_t)s4 * Tin + g)) : x0[s4];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gj = g + j;
                bool o = gj >= -padL && gj < hi_lim;
                int src = gj < 0 ? -gj : gj;
                src = src >= Leff ? refl - src : src;
                if (pad_zero) { src = gj; o = o && gj >= 0; }
                o = o && src >= 0 && src < Tin;          // zero padding / zero-extension of short inputs (conv.py:89-93)
                src = o ? src : 0;
                ok[j] = o;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    x0[s4][j] = r0[(size_t)s4 * Tin + src];
                    x1[s4][j] = s1.ptr ? r1[(size_t)s4 * Tin + src] : 0.f;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (4 * q + j >= Tp) break;
            f32x4 y;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) y[s4] = ok[j] ? act(x0[s4][j], x1[s4][j], s4) : 0.f;
            orow[4 * q + j] = y;
        }
    }
}

bool conv_xq_ok(int Cin, int CC, int k, int stride, int dil, int BM, int BN, int row) {
    static const int xq_env = ab_knob("FC_XQ", 1);          // FC_XQ=0: register staging everywhere (A / B aid)
    if (!xq_env || !conv_quad(CC) || Cin % 4 != 0 || Cin % CC != 0) return false;
    const int slabW = (BN - 1) * stride + (k - 1) * dil + 1;
    const int rowStride = row ? ((slabW + 3) & ~3) : ceil_div(slabW, stride) * stride;
    (void)BM;
    return ceil_div((CC / 4) * rowStride, 256) <= 6;
}
size_t conv_xq_floats(int B, int Cin, int Tp, int BN, int stride, int k, int dil) {
    // the DMA of the last N tile reads a whole slab (and the pad lanes of its last round read row 0 of the chunk): slack of one slab + one
    // round of pieces behind the last row
    return ((size_t)B * (Cin / 4) * (size_t)Tp + (size_t)BN * stride + (size_t)k * dil + 64 + 256) * 4;
}
hipError_t launch_combine_xq(const Src& s0, const Src& s1, int elu, float alpha, int B, int C, int Tin, int padL, int padR, int pad_zero,
                             float* xq, hipStream_t st) {
    if (C % 4 != 0 || B <= 0 || Tin <= 0 || s0.div) return hipErrorInvalidValue;
    const int Tp = padL + Tin + padR;
    const int maxpad = padL > padR ? padL : padR;
    const int Leff = Tin > maxpad ? Tin : maxpad + 1;
    int gx = ceil_div(ceil_div(Tp, 4), 256);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(combine_xq_kernel, dim3(gx, C / 4, B), dim3(256), 0, st, s0, s1, elu, alpha, C, Tin, padL, padR, pad_zero, Leff, xq);
    return hipGetLastError();
}

// volume = sqrt(mean(mono^2)); scale = 1e-8 + volume, mono = the channel mean        (Encodec._encode_frame codec_basic.py:366-371)
// wav [B][C][T], C = 1 or 2 (stereo: mono = (left + right) / 2, the fp32 sum torch's x.mean(dim=1) forms, halved exactly)
template <int C>
__global__ __launch_bounds__(1024) void volume_kernel(const float* __restrict__ wav, int T, float* __restrict__ scale) {
    __shared__ double sh[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = wav + (size_t)b * C * T;
    // 16-byte loads (dword alignment only: T need not be a multiple of 4) and four independent fp64 chains per thread:
    // one dependent add chain per element kept this 10 MB reduction at 70 us
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int T4 = T >> 2;
    for (int q = tid; q < T4; q += 1024) {
        f32x4 v = *(const f32x4u*)(x + 4 * q);
        if (C == 2) v = (v + (f32x4)(*(const f32x4u*)(x + T + 4 * q))) * 0.5f;
        s0 += (double)(v[0] * v[0]); s1 += (double)(v[1] * v[1]); s2 += (double)(v[2] * v[2]); s3 += (double)(v[3] * v[3]);
    }
    for (int t = 4 * T4 + tid; t < T; t += 1024) {
        float v = x[t];
        if (C == 2) v = (v + x[T + t]) * 0.5f;
        s0 += (double)(v * v);
    }
    const double s = (s0 + s1) + (s2 + s3);
    sh[tid] = s;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    if (tid == 0) scale[b] = 1e-8f + sqrtf((float)(sh[0] / (double)T));
}

hipError_t launch_volume(const float* wav, int B, int C, int T, float* scale, hipStream_t st) {
    if (C == 1) hipLaunchKernelGGL(volume_kernel<1>, dim3(B), dim3(1024), 0, st, wav, T, scale);
    else if (C == 2) hipLaunchKernelGGL(volume_kernel<2>, dim3(B), dim3(1024), 0, st, wav, T, scale);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void transpose_btd_kernel(const float* __restrict__ in, int T, int D, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, d = d0 + tx;
        tile[i][tx] = (t < T && d < D) ? in[((size_t)b * T + t) * D + d] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int d = d0 + i, t = t0 + tx;
        if (t < T && d < D) out[((size_t)b * D + d) * T + t] = tile[tx][i];
    }
}

hipError_t launch_transpose_btd(const float* in, int B, int T, int D, float* out, hipStream_t st) {
    hipLaunchKernelGGL(transpose_btd_kernel, dim3(ceil_div(T, 32), ceil_div(D, 32), B), dim3(256), 0, st, in, T, D, out);
    return hipGetLastError();
}

// =================================================================================================
// 4. Residual vector quantiser: all n_q stages fused, 16 rows per workgroup
//    (DistributedResidualVectorQuantization.forward ddp_core_vq.py:367-418, eval branch;
//     EuclideanCodebook.quantize :180-188; dequantize :190-192)
//
//    stage i:  dist[row][k] = -(( |x|^2 - (2x).e_k ) + |e_k|^2)     idx = argmax_k, FIRST max wins
//              q = E_i[idx];  residual -= q;  out += q
//
//    Arithmetic order (restated bit-for-bit by oracle/c/rvq_oracle.c):
//      |x|^2   : 4 partial chains over d in [j*D/4,(j+1)*D/4), s = s + x*x with the square rounded
//                separately (torch: x.pow(2).sum(1)), combined (p0+p1)+(p2+p3);
//      (2x).e  : one MFMA accumulator per (row, code): fmaf chain from 0 over d in the order
//                d = 16q + 4g + j  for q = 0..D/16-1, j = 0..3, g = 0..3 (innermost);
//      |e|^2   : precomputed at load time (engine.hip), sequential d = 0..D-1, squares rounded separately.
// =================================================================================================
// RS = row sets of 16 rows per workgroup.  The stage is L2-bandwidth bound when one codebook load feeds only 16 rows
// (every workgroup streams the whole 512 KiB stage codebook: 250 workgroups x 512 KiB = 128 MiB per stage at config B);
// with RS = 2 the same registers feed two independent 16-row MFMA accumulators (same arithmetic per row).
// Q0: the stage-0 source-row table `src0` is honoured (quantizer_conf.q0_ds_ratio > 1); a template parameter so that the benchmark's
// instantiations keep their register allocation (183 registers, no spill).
template <int D, int RS, bool Q0 = false>
__global__ __launch_bounds__(512) void rvq_encode_kernel(const float* __restrict__ x, int N, int K, int nq,
                                                         const float* __restrict__ cb, const float* __restrict__ cbf,
                                                         const float* __restrict__ enorm,
                                                         int64_t* __restrict__ codes, float* __restrict__ quant,
                                                         float* __restrict__ quant_bdt, float* __restrict__ subq, int Tf,
                                                         int ablate, const int* __restrict__ src0) {
    constexpr int NQ4 = D / 16;
    constexpr int ROWS = 16 * RS;
    // residual rows, padded by 4 floats: the |x|^2 chains read R[r][32 j + i] from 64 lanes (r, j) at once -- with a row stride of
    // D every lane hit the same bank (32-way conflict, ~1 us per stage); stride D + 4 spreads the 8 rows of a lane group over 8 banks
    __shared__ __attribute__((aligned(16))) float R[ROWS][D + 4];
    constexpr int NEL = (ROWS * D + 511) / 512;                // elements of the row set owned by one thread: e = tid + 512 * it
    float qreg[NEL];                                    // running sum of the selected code rows (was an LDS array)
    __shared__ float xn[ROWS];
    __shared__ float bestv[8][ROWS];
    __shared__ int besti[8][ROWS];
    __shared__ int sel[ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int row0 = blockIdx.x * ROWS;

    for (int e = tid; e < ROWS * D; e += 512) {
        const int r = e / D, d = e - r * D;
        const int n = row0 + r;
        // src0 (quantizer_conf.q0_ds_ratio > 1, ddp_core_vq.py:396-404): stage 0 quantises the row its frame copies from the
        // half-rate sequence; the residual update of that stage starts again from the row itself (below)
        R[r][d] = n < N ? x[(size_t)(Q0 ? src0[n] : n) * D + d] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NEL; ++it) qreg[it] = 0.f;
    const int codes_per_wave = (K >> 3) < 16 ? 16 : (K >> 3);   // 8 waves: two per SIMD hide the codebook-row load latency
    const bool wactive = wid * codes_per_wave < K;            // small codebooks keep only K/16 waves busy

    constexpr bool DEEP = RS == 1;                                // one row set: codebook tiles two ahead (the two-row-set form has twice the MFMAs per tile)
    f32x4 b0[NQ4];                                               // first codebook fragment of the stage (requested one stage ahead)
    f32x4 b1p[DEEP ? NQ4 : 1];                                   // DEEP: the second one
    for (int i = 0; i < nq; ++i) {
        __syncthreads();
        if (tid < 4 * ROWS) {   // |x|^2
            const int r = tid >> 2, j = tid & 3;
            float s = 0.f;
#pragma unroll 8
            for (int d = j * (D / 4); d < (j + 1) * (D / 4); ++d) {
                const float v = R[r][d];
                s = __fadd_rn(s, __fmul_rn(v, v));
            }
            const float s_pair = __fadd_rn(s, __shfl_xor(s, 1, 64));          // p0+p1 | p2+p3
            const float s_all = __fadd_rn(s_pair, __shfl_xor(s_pair, 2, 64));  // (p0+p1)+(p2+p3)
            if (j == 0) xn[r] = s_all;
        }
        f32x4 a4[RS][NQ4];
#pragma unroll
        for (int s2 = 0; s2 < RS; ++s2)
#pragma unroll
            for (int q = 0; q < NQ4; ++q) {
                const f32x4 v = *(const f32x4*)&R[16 * s2 + r16][16 * q + 4 * g];
                a4[s2][q] = v + v;   // 2*x, exact
            }
        __syncthreads();
        float best[RS][4];
        int bidx[RS][4];
        float xr[RS][4];
#pragma unroll
        for (int s2 = 0; s2 < RS; ++s2)
#pragma unroll
            for (int r = 0; r < 4; ++r) { best[s2][r] = -INFINITY; bidx[s2][r] = 0x7fffffff; xr[s2][r] = xn[16 * s2 + 4 * g + r]; }
        const float* cbi = cb + (size_t)i * K * D;
        // One 16-code tile per trip, software pipelined: the codebook fragment of tile n+1 is requested before the 4*NQ4
        // MFMAs of tile n are issued (two register buffers), so the L2 latency of the stream hides behind the matrix work;
        // the second wave on the SIMD fills the dependent-MFMA bubbles.  Each (row, code) chain keeps its own d order.
        const int ntile = wactive ? codes_per_wave >> 4 : 0;
        const int code0 = wid * codes_per_wave;
        auto frag_ptr = [&](int stage, int t) __attribute__((always_inline)) {
            const int code = code0 + 16 * t;
            return cbf ? cbf + ((size_t)stage * K + code) * D + lane * 4 : cb + ((size_t)stage * K + code + r16) * D + 4 * g;
        };
        const int qstep = cbf ? 256 : 16;
        auto load_tile = [&](int stage, int t, f32x4 (&bq)[NQ4]) __attribute__((always_inline)) {
            const float* e0 = frag_ptr(stage, t);
#pragma unroll
            for (int q = 0; q < NQ4; ++q) bq[q] = FC_ABL(ablate, 2) ? (f32x4){1.f, 1.f, 1.f, 1.f} : *(const f32x4*)(e0 + qstep * q);
        };
        auto do_tile = [&](int t, const f32x4 (&bq)[NQ4]) __attribute__((always_inline)) {
            const int code = code0 + 16 * t + r16;
            const float en = enorm[(size_t)i * K + code];
            f32x4 acc[RS];
#pragma unroll
            for (int s2 = 0; s2 < RS; ++s2) acc[s2] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!FC_ABL(ablate, 1)) {
#pragma unroll
                for (int q = 0; q < NQ4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int s2 = 0; s2 < RS; ++s2)
                            acc[s2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s2][q][j], bq[q][j], acc[s2], 0, 0, 0);
            }
#pragma unroll
            for (int s2 = 0; s2 < RS; ++s2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dist = -__fadd_rn(__fsub_rn(xr[s2][r], acc[s2][r]), en);
                    if (dist > best[s2][r]) { best[s2][r] = dist; bidx[s2][r] = code; }
                }
        };
        if (ntile > 0) {
            if constexpr (DEEP) {
                // Round 5: codebook tiles are requested TWO tiles ahead (three register buffers).  With one tile ahead the L2 round trip of a
                // fragment (~1 us under this load) is as long as the 2 x 0.43 us of MFMAs of the two waves that share a SIMD.  Measured: 525 -> 517 us
                // (of the 80 us the codebook loads cost on top of the MFMAs -- tools/ablate_rvq.py -- most is their issue time, not latency)
                f32x4 b2[NQ4];
                if (i == 0) { load_tile(0, 0, b0); if (ntile > 1) load_tile(0, 1, b1p); }
                int t = 0;
                for (; t + 3 <= ntile; t += 3) {
                    load_tile(i, t + 2, b2);
                    do_tile(t, b0);
                    if (t + 3 < ntile) load_tile(i, t + 3, b0);
                    do_tile(t + 1, b1p);
                    if (t + 4 < ntile) load_tile(i, t + 4, b1p);
                    do_tile(t + 2, b2);
                }
                if (t < ntile) do_tile(t, b0);                 // 0, 1 or 2 tiles left: already requested into b0 / b1p
                if (t + 1 < ntile) do_tile(t + 1, b1p);
                // the next stage's first fragments do not depend on this stage's result: requested now, the arg-max / residual-update tail
                // of this stage hides their latency
                if (i + 1 < nq) { load_tile(i + 1, 0, b0); if (ntile > 1) load_tile(i + 1, 1, b1p); }
            } else {
                f32x4 b1[NQ4];
                if (i == 0) load_tile(0, 0, b0);
                int t = 0;
                for (; t + 2 <= ntile; t += 2) {
                    load_tile(i, t + 1, b1);
                    do_tile(t, b0);
                    if (t + 2 < ntile) load_tile(i, t + 2, b0);
                    do_tile(t + 1, b1);
                }
                if (t < ntile) do_tile(t, b0);
                // the next stage's first fragment does not depend on this stage's result: request it now, the arg-max /
                // residual-update tail of this stage hides its latency
                if (i + 1 < nq) load_tile(i + 1, 0, b0);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < RS; ++s2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    const float ov = __shfl_xor(best[s2][r], o, 64);
                    const int oi = __shfl_xor(bidx[s2][r], o, 64);
                    if (ov > best[s2][r] || (ov == best[s2][r] && oi < bidx[s2][r])) { best[s2][r] = ov; bidx[s2][r] = oi; }
                }
                if (r16 == 0) { bestv[wid][16 * s2 + 4 * g + r] = best[s2][r]; besti[wid][16 * s2 + 4 * g + r] = bidx[s2][r]; }
            }
        }
        __syncthreads();
        if (tid < ROWS) {
            float bv = bestv[0][tid];
            int bi = besti[0][tid];
            for (int w = 1; w < 8; ++w) {
                const float ov = bestv[w][tid];
                const int oi = besti[w][tid];
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (bi < 0 || bi >= K) bi = 0;   // all-NaN row: torch would return an index too; stay in range
            sel[tid] = bi;
            if (row0 + tid < N) codes[(size_t)i * N + row0 + tid] = (int64_t)bi;
        }
        __syncthreads();
        // gather the selected code rows: all loads first (they are independent), then the LDS / register updates
        float qv[NEL];
#pragma unroll
        for (int it = 0; it < NEL; ++it) {
            const int e = tid + 512 * it, r = e / D, d = e - r * D;
            qv[it] = 0.f;
            if (ROWS * D % 512 == 0 || e < ROWS * D) qv[it] = FC_ABL(ablate, 8) ? 0.001f : cbi[(size_t)sel[r] * D + d];
        }
#pragma unroll
        for (int it = 0; it < NEL; ++it) {
            const int e = tid + 512 * it, r = e / D, d = e - r * D;
            if (ROWS * D % 512 != 0 && e >= ROWS * D) continue;
            const int n = row0 + r;
            float res = R[r][d];
            if (Q0 && i == 0 && n < N) res = x[(size_t)n * D + d];
            R[r][d] = res - qv[it];
            qreg[it] = qreg[it] + qv[it];
            if (subq && n < N && !FC_ABL(ablate, 4)) {
                const int bb = n / Tf, t = n - bb * Tf;
                const int Bn = N / Tf;
                subq[(((size_t)i * Bn + bb) * D + d) * Tf + t] = qv[it];
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NEL; ++it) {
        const int e = tid + 512 * it, r = e / D, d = e - r * D;
        const int n = row0 + r;
        if (n >= N || (ROWS * D % 512 != 0 && e >= ROWS * D)) continue;
        const float v = qreg[it];
        if (quant) quant[(size_t)n * D + d] = v;
        if (quant_bdt) {
            const int bb = n / Tf, t = n - bb * Tf;
            quant_bdt[((size_t)bb * D + d) * Tf + t] = v;
        }
    }
}

// Wide codebooks (D = 512, the SoundStream recipe): same arithmetic and the same argmax / update flow as rvq_encode_kernel,
// but a (row, code) chain is fed in chunks of 256 dims -- the 2x fragment of the row (re-read from LDS per chunk) and the two
// in-flight codebook fragments then fit the register file -- and the running quantised sum lives in registers instead of LDS.
template <int D, bool Q0 = false>
__global__ __launch_bounds__(512) void rvq_encode_wide_kernel(const float* __restrict__ x, int N, int K, int nq,
                                                              const float* __restrict__ cb, const float* __restrict__ cbf,
                                                              const float* __restrict__ enorm,
                                                              int64_t* __restrict__ codes, float* __restrict__ quant,
                                                              float* __restrict__ quant_bdt, float* __restrict__ subq, int Tf,
                                                              const int* __restrict__ src0) {
    static_assert(D % 256 == 0, "chunks of 256 dims");
    constexpr int NC = D / 256, CQ = 16, NEL = 16 * D / 512;
    __shared__ __attribute__((aligned(16))) float R[16][D];
    __shared__ float xn[16];
    __shared__ float bestv[8][16];
    __shared__ int besti[8][16];
    __shared__ int sel[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int row0 = blockIdx.x * 16;
    float qreg[NEL];                                     // running sum of the selected code rows, element e = tid + 512*i
#pragma unroll
    for (int i2 = 0; i2 < NEL; ++i2) {
        const int e = tid + 512 * i2, r = e / D, d = e - r * D, n = row0 + r;
        R[r][d] = n < N ? x[(size_t)(Q0 ? src0[n] : n) * D + d] : 0.f;       // Q0 / src0: see rvq_encode_kernel
        qreg[i2] = 0.f;
    }
    const int codes_per_wave = (K >> 3) < 16 ? 16 : (K >> 3);
    const bool wactive = wid * codes_per_wave < K;
    const int ntile = wactive ? codes_per_wave >> 4 : 0, nunit = ntile * NC;
    const int code0 = wid * codes_per_wave;
    for (int i = 0; i < nq; ++i) {
        __syncthreads();
        if (tid < 64) {   // |x|^2, same chains as the narrow kernel
            const int r = tid >> 2, j = tid & 3;
            float s = 0.f;
#pragma unroll 8
            for (int d = j * (D / 4); d < (j + 1) * (D / 4); ++d) {
                const float v = R[r][d];
                s = __fadd_rn(s, __fmul_rn(v, v));
            }
            const float s_pair = __fadd_rn(s, __shfl_xor(s, 1, 64));
            const float s_all = __fadd_rn(s_pair, __shfl_xor(s_pair, 2, 64));
            if (j == 0) xn[r] = s_all;
        }
        __syncthreads();
        float best[4], xr[4];
        int bidx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { best[r] = -INFINITY; bidx[r] = 0x7fffffff; xr[r] = xn[4 * g + r]; }
        const float* cbi = cb + (size_t)i * K * D;
        auto load_unit = [&](int u, f32x4 (&bq)[CQ]) __attribute__((always_inline)) {
            const int t = u / NC, c = u - t * NC, code = code0 + 16 * t;
            const float* e0 = cbf ? cbf + ((size_t)i * K + code) * D + (size_t)c * CQ * 256 + lane * 4
                                  : cbi + (size_t)(code + r16) * D + 256 * c + 4 * g;
            const int qstep = cbf ? 256 : 16;
#pragma unroll
            for (int q = 0; q < CQ; ++q) bq[q] = *(const f32x4*)(e0 + qstep * q);
        };
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        auto do_unit = [&](int u, const f32x4 (&bq)[CQ]) __attribute__((always_inline)) {
            const int t = u / NC, c = u - t * NC, code = code0 + 16 * t + r16;
            if (c == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
                const f32x4 v = *(const f32x4*)&R[r16][256 * c + 16 * q + 4 * g];
                const f32x4 a = v + v;                                  // 2*x, exact
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bq[q][j], acc, 0, 0, 0);
            }
            if (c == NC - 1) {
                const float en = enorm[(size_t)i * K + code];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dist = -__fadd_rn(__fsub_rn(xr[r], acc[r]), en);
                    if (dist > best[r]) { best[r] = dist; bidx[r] = code; }
                }
            }
        };
        if (nunit > 0) {
            f32x4 b0[CQ], b1[CQ];
            load_unit(0, b0);
            int u = 0;
            for (; u + 2 <= nunit; u += 2) {
                load_unit(u + 1, b1);
                do_unit(u, b0);
                if (u + 2 < nunit) load_unit(u + 2, b0);
                do_unit(u + 1, b1);
            }
            if (u < nunit) do_unit(u, b0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const float ov = __shfl_xor(best[r], o, 64);
                const int oi = __shfl_xor(bidx[r], o, 64);
                if (ov > best[r] || (ov == best[r] && oi < bidx[r])) { best[r] = ov; bidx[r] = oi; }
            }
            if (r16 == 0) { bestv[wid][4 * g + r] = best[r]; besti[wid][4 * g + r] = bidx[r]; }
        }
        __syncthreads();
        if (tid < 16) {
            float bv = bestv[0][tid];
            int bi = besti[0][tid];
            for (int w = 1; w < 8; ++w) {
                const float ov = bestv[w][tid];
                const int oi = besti[w][tid];
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (bi < 0 || bi >= K) bi = 0;
            sel[tid] = bi;
            if (row0 + tid < N) codes[(size_t)i * N + row0 + tid] = (int64_t)bi;
        }
        __syncthreads();
#pragma unroll
        for (int i2 = 0; i2 < NEL; ++i2) {
            const int e = tid + 512 * i2, r = e / D, d = e - r * D, n = row0 + r;
            const float qv = cbi[(size_t)sel[r] * D + d];
            float res = R[r][d];
            if (Q0 && i == 0 && n < N) res = x[(size_t)n * D + d];
            R[r][d] = res - qv;
            qreg[i2] = qreg[i2] + qv;
            if (subq && n < N) {
                const int bb = n / Tf, t = n - bb * Tf, Bn = N / Tf;
                subq[(((size_t)i * Bn + bb) * D + d) * Tf + t] = qv;
            }
        }
    }
#pragma unroll
    for (int i2 = 0; i2 < NEL; ++i2) {
        const int e = tid + 512 * i2, r = e / D, d = e - r * D, n = row0 + r;
        if (n >= N) continue;
        if (quant) quant[(size_t)n * D + d] = qreg[i2];
        if (quant_bdt) {
            const int bb = n / Tf, t = n - bb * Tf;
            quant_bdt[((size_t)bb * D + d) * Tf + t] = qreg[i2];
        }
    }
}

// quantizer_conf.q0_ds_ratio > 1 (ddp_core_vq.py:396-404): map[b * Tf + t] = b * Tf + q0_source_frame(t, Tf)   (kernels.h)
__global__ __launch_bounds__(256) void q0_map_kernel(int* __restrict__ map, int Tf) {
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (t < Tf) map[(size_t)b * Tf + t] = b * Tf + q0_source_frame(t, Tf);
}

hipError_t launch_q0_map(int* map, int B, int Tf, hipStream_t st) {
    if (B <= 0 || Tf < 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(q0_map_kernel, dim3(ceil_div(Tf, 256), B), dim3(256), 0, st, map, Tf);
    return hipGetLastError();
}

hipError_t launch_rvq_encode(const float* x, int N, int D, int K, int nq, const float* cb, const float* cb_frag,
                             const float* enorm, int64_t* codes, float* quant, float* quant_bdt, float* subq, int Tf, hipStream_t st,
                             const int* src0) {
    if (N <= 0) return hipSuccess;
    if (K % 16 != 0 || (K > 128 && K % 128 != 0)) return hipErrorInvalidValue;
    if (D == 512) {
        if (src0) hipLaunchKernelGGL((rvq_encode_wide_kernel<512, true>), dim3(ceil_div(N, 16)), dim3(512), 0, st, x, N, K, nq, cb, cb_frag, enorm,
                                     codes, quant, quant_bdt, subq, Tf, src0);
        else hipLaunchKernelGGL((rvq_encode_wide_kernel<512>), dim3(ceil_div(N, 16)), dim3(512), 0, st, x, N, K, nq, cb, cb_frag, enorm, codes,
                           quant, quant_bdt, subq, Tf, src0);
        return hipGetLastError();
    }
    // two row sets per workgroup once there are enough rows to keep ~half the CUs busy that way (L2 traffic halves)
    static const int ablate = ab_knob("FC_ABLATE_RVQ", 0);
    // Round 4 (tools/rvq_scaling.py): the kernel's time is linear in the workgroups per CU (one 8-wave workgroup per CU, 183 registers).
    // Up to 16 rows per CU the 16-row form is the fastest (497 vs 807 us at 4 000 rows); beyond that two row sets per workgroup -- the
    // codebook fragments of a stage are loaded once for both, 32 rows per MFMA pass -- take 413 us per 32 rows per CU against 2 x 489:
    // 846 vs 974 us at 8 000 rows, 1 613 vs 1 943 at 16 000.  FC_RVQ_TWO=0 / 1 forces one form (A / B runs).
    static const int two_env = ab_knob("FC_RVQ_TWO", -1);
    // CU count of the CURRENT device, looked up per call from a table filled once per device (thread-safe: function-local static of a
    // type with a constructor; ADVICE r4: the round-4 form cached whichever device was current at the first call, unsynchronised)
    struct CuTable {
        int n[64];
        CuTable() {
            int nd = 0;
            if (hipGetDeviceCount(&nd) != hipSuccess) nd = 0;
            for (int d = 0; d < 64; ++d) {
                hipDeviceProp_t prop;
                n[d] = (d < nd && hipGetDeviceProperties(&prop, d) == hipSuccess) ? prop.multiProcessorCount : 256;
            }
        }
    };
    static const CuTable cu_table;
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    const int n_cus = cu_table.n[cur_dev & 63];
    const bool two = !src0 && D <= 128 && (two_env >= 0 ? (two_env != 0 && N >= 2048) : N > 16 * n_cus);
    dim3 grid(ceil_div(N, two ? 32 : 16)), block(512);
#define FC_RVQ_CASE(DD)                                                                                            \
    case DD:                                                                                                       \
        if (two) hipLaunchKernelGGL((rvq_encode_kernel<(DD <= 128 ? DD : 16), 2>), grid, block, 0, st, x, N, K, nq, cb, cb_frag, enorm, codes, quant, \
                                    quant_bdt, subq, Tf, ablate, src0);                                            \
        else if (src0) hipLaunchKernelGGL((rvq_encode_kernel<DD, 1, true>), grid, block, 0, st, x, N, K, nq, cb, cb_frag, enorm, codes, quant, quant_bdt, \
                                subq, Tf, ablate, src0);                                                           \
        else hipLaunchKernelGGL((rvq_encode_kernel<DD, 1>), grid, block, 0, st, x, N, K, nq, cb, cb_frag, enorm, codes, quant, quant_bdt, \
                                subq, Tf, ablate, src0);                                                           \
        break;
    switch (D) {
        FC_RVQ_CASE(16)
        FC_RVQ_CASE(32)
        FC_RVQ_CASE(64)
        FC_RVQ_CASE(128)
        FC_RVQ_CASE(256)
        default: return hipErrorInvalidValue;
    }
#undef FC_RVQ_CASE
    return hipGetLastError();
}

// DRVQ.decode (ddp_core_vq.py:442-453): out = ((0 + E_0[i0]) + E_1[i1]) + ...
__global__ __launch_bounds__(256) void rvq_decode_kernel(const int64_t* __restrict__ codes, int Tf, int nq, int D, int K,
                                                         const float* __restrict__ cb, float* __restrict__ emb,
                                                         float* __restrict__ emb_bdt, unsigned* __restrict__ status) {
    const int n = blockIdx.x;   // row = b*Tf + t
    const int b = n / Tf, t = n - b * Tf;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < nq; ++i) {
            long long idx = codes[(size_t)n * nq + i];
            // F.embedding raises on an index outside [0, K) (ddp_core_vq.py:191): never read out of bounds, but make the
            // corrupt token loud -- the engine status word reports it (fc_engine_status)
            if (idx < 0 || idx >= K) {
                if (status && threadIdx.x == 0) *(volatile unsigned*)(status + FC_STATUS_BAD_CODE) = 1u;
                idx = idx < 0 ? 0 : K - 1;
            }
            s = s + cb[((size_t)i * K + idx) * D + d];
        }
        if (emb) emb[(size_t)n * D + d] = s;
        if (emb_bdt) emb_bdt[((size_t)b * D + d) * Tf + t] = s;
    }
}

hipError_t launch_rvq_decode(const int64_t* codes, int B, int Tf, int nq, int D, int K, const float* cb, float* emb,
                             float* emb_bdt, unsigned* status, hipStream_t st) {
    if (B * Tf <= 0) return hipSuccess;
    const int threads = D >= 256 ? 256 : (D >= 128 ? 128 : 64);
    hipLaunchKernelGGL(rvq_decode_kernel, dim3(B * Tf), dim3(threads), 0, st, codes, Tf, nq, D, K, cb, emb, emb_bdt, status);
    return hipGetLastError();
}

// =================================================================================================
// 5. LSTM (nn.LSTM inside SLSTM, lstm.py:12-28).  Workgroup j owns the 16 gate rows {i,f,g,o} x 4 hidden units (rows
//    permuted at load time), split-K over its waves, gates + state update fused.
//    gates = (W_hh h_{t-1}) + xproj_t,  xproj = W_ih x_t + b_ih + b_hh computed for all t at once by the conv kernel (k = 1).
// =================================================================================================
// Gate nonlinearities on the hardware exp2 / rcp (1 ulp each): sigmoid(v) = 1 / (1 + 2^(-v log2 e)), tanh(v) = 2 sigmoid(2v) - 1.
// Absolute error <= ~2e-7 (the libm versions are ~40 and ~60 VALU instructions and sit on the recurrence's critical path);
// measured end to end: every golden index still bit-exact, LSTM outputs within 1e-5 of torch (tests/test_gpu_parity.py).
__device__ __forceinline__ float sigmoid_f(float v) {
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * v);      // +inf for very negative v -> rcp gives 0
    return __builtin_amdgcn_rcpf(1.f + e);
}
__device__ __forceinline__ float tanh_f(float v) {
    const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * v);
    return fmaf(2.f, __builtin_amdgcn_rcpf(1.f + e), -1.f);
}
// c' = f * c + i * g with ONE explicit rounding order -- fma(f, c, round(i * g)) -- in every LSTM kernel: left to hipcc's contraction the
// same expression became v_mul + v_fmac in one kernel and v_pk_mul + v_add in another (one ulp apart; the persistent and the per-step
// forms are tested for bit equality)
__device__ __forceinline__ float lstm_cell(float gf, float c, float gi, float gg) { return __fmaf_rn(gf, c, __fmul_rn(gi, gg)); }

// -------------------------------------------------------------------------------------------------
// Layer-wavefront LSTM step: launch s advances layer l by its timestep t = s - l for ALL layers at once
// (T + L - 1 launches instead of L*T, and no separate x-projection GEMM for layers >= 1, whose input
// h_{l-1}(t) is consumed straight from the previous launch):
//   layer 0 : gates = xproj[t] + W_hh0 . h0(t-1)
//   layer l : gates = bias_l + [W_ih_l | W_hh_l] . [h_{l-1}(t) ; h_l(t-1)]
// hidden states ping-pong on the parity of their own timestep: h[l][t & 1].
// -------------------------------------------------------------------------------------------------
struct LstmWaveArgs {
    const float* w[FC_LSTM_MAX_LAYERS];
    const float* bias[FC_LSTM_MAX_LAYERS];
    const float* xproj;
    float* h;
    float* c;
    float* y;
    int B, H, T, L, s, KS;
};

template <int NS>
__global__ __launch_bounds__(256) void lstm_wave_kernel(const LstmWaveArgs p) {
    __shared__ f32x4 red[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int H = p.H, B = p.B;
    const int nblk = H >> 2;
    const int layer = blockIdx.x / nblk, blk = blockIdx.x - layer * nblk;
    const int t = p.s - layer;
    if (t < 0 || t >= p.T) return;
    const int kslice = H / p.KS;
    const int nsteps = kslice >> 4;
    const bool active = wid < p.KS;
    const int nseg = layer == 0 ? 1 : 2;
    const int wstride = layer == 0 ? H : 2 * H;
    const float* wbase = p.w[layer] + ((size_t)blk * 16 + r16) * wstride + (active ? wid : 0) * kslice + 4 * g;
    const size_t BH = (size_t)B * H;
    float* hl = p.h + (size_t)layer * 2 * BH;
    const float* h_own_prev = hl + (size_t)((t + 1) & 1) * BH;                  // h_l(t-1): parity (t-1)&1
    const float* h_below = layer ? p.h + (size_t)(layer - 1) * 2 * BH + (size_t)(t & 1) * BH : nullptr;   // h_{l-1}(t)
    float* h_out = hl + (size_t)(t & 1) * BH;
    float* cl = p.c + (size_t)layer * BH;
    constexpr int NA = NS > 0 ? NS : 1;
    const int nbt = (B + 15) >> 4;
    for (int nb = 0; nb < nbt; ++nb) {
        const int brow = nb * 16 + r16;
        const bool bvalid = brow < B;
        const size_t ci = (size_t)(bvalid ? brow : 0) * H + (size_t)blk * 4 + g;
        f32x4 xp = {0.f, 0.f, 0.f, 0.f};
        float cprev = 0.f;
        if (wid == 0) {
            if (layer == 0) xp = *(const f32x4*)(p.xproj + ((size_t)t * B + (bvalid ? brow : 0)) * 4 * H + (size_t)blk * 16 + 4 * g);
            else xp = *(const f32x4*)(p.bias[layer] + (size_t)blk * 16 + 4 * g);
            cprev = cl[ci];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int seg = 0; seg < nseg; ++seg) {
            const float* wrow = wbase + (nseg == 2 ? seg * H : 0);
            const float* hsrc = (nseg == 2 && seg == 0) ? h_below : h_own_prev;
            const float* hrow = hsrc + (size_t)(bvalid ? brow : 0) * H + (active ? wid : 0) * kslice + 4 * g;
            if (NS > 0) {
                f32x4 a4[NA], b4[NA];
#pragma unroll
                for (int q = 0; q < NA; ++q) {
                    a4[q] = *(const f32x4*)(wrow + 16 * q);
                    b4[q] = *(const f32x4*)(hrow + 16 * q);
                    if (!bvalid) b4[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int q = 0; q < NA; ++q) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (q & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q][j], b4[q][j], acc1, 0, 0, 0);
                        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q][j], b4[q][j], acc, 0, 0, 0);
                    }
                }
            } else if (active) {
                for (int q = 0; q < nsteps; ++q) {
                    const f32x4 av = *(const f32x4*)(wrow + 16 * q);
                    f32x4 bv = *(const f32x4*)(hrow + 16 * q);
                    if (!bvalid) bv = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
                }
            }
        }
        acc = acc + acc1;
        if (!active) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        red[wid][lane] = acc;
        __syncthreads();
        if (wid == 0) {
            f32x4 sgate = red[0][lane];
            for (int w = 1; w < p.KS; ++w) sgate = sgate + red[w][lane];
            if (bvalid) {
                const float gi = sigmoid_f(sgate[0] + xp[0]);
                const float gf = sigmoid_f(sgate[1] + xp[1]);
                const float gg = tanh_f(sgate[2] + xp[2]);
                const float go = sigmoid_f(sgate[3] + xp[3]);
                const float cn = lstm_cell(gf, cprev, gi, gg);
                const float hn = go * tanh_f(cn);
                cl[ci] = cn;
                h_out[ci] = hn;
                if (layer == p.L - 1) p.y[ci * p.T + t] = hn;
            }
        }
        __syncthreads();
    }
}

hipError_t launch_lstm_wave(const float* const* w, const float* const* bias, const float* xproj, float* h, float* c,
                            float* y, int B, int H, int T, int L, int s, hipStream_t st) {
    if (H % 16 != 0 || L < 1 || L > FC_LSTM_MAX_LAYERS) return hipErrorInvalidValue;
    LstmWaveArgs a;
    for (int l = 0; l < FC_LSTM_MAX_LAYERS; ++l) { a.w[l] = l < L ? w[l] : nullptr; a.bias[l] = l < L ? bias[l] : nullptr; }
    a.xproj = xproj; a.h = h; a.c = c; a.y = y; a.B = B; a.H = H; a.T = T; a.L = L; a.s = s;
    int KS = 4;
    while (KS > 1 && (H % (16 * KS)) != 0) KS >>= 1;
    a.KS = KS;
    const int ns = H / (16 * KS);
    dim3 grid(L * (H / 4)), block(256);
    switch (ns) {
        case 1: hipLaunchKernelGGL(lstm_wave_kernel<1>, grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL(lstm_wave_kernel<2>, grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL(lstm_wave_kernel<4>, grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL(lstm_wave_kernel<8>, grid, block, 0, st, a); break;
        case 16: hipLaunchKernelGGL(lstm_wave_kernel<16>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(lstm_wave_kernel<0>, grid, block, 0, st, a); break;
    }
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Persistent 2-layer LSTM: ONE launch for the whole recurrence.  Workgroup j keeps the 16 gate rows of layer 0
// (W_hh0) and of layer 1 ([W_ih1 | W_hh1]) for its 4 hidden units in REGISTERS for all T steps (192 KiB per CU,
// one 4-wave workgroup per CU owns the whole 512-register file), cell states stay in registers too.
//
// Hidden-state exchange.  Every wavefront step writes h0(s) / h1(s-1) into a HISTORY buffer that is never
// overwritten inside one launch ([T+1][B][H] per layer, slot 0 = zeros).  Stores are write-through (sc1), so when a
// wave's vmcnt drains its values are in memory; the consumers read slot s only after the grid barrier of step s-1 and
// nobody has touched those addresses before in this launch, so no cache on the chip can hold a stale copy of them:
// the loads are PLAIN loads, the first workgroup of an XCD to touch a line pulls it over the fabric and the other 31
// hit that XCD's L2 (with double-buffered h and sc1 loads every workgroup fetched all 128 KiB over the fabric every
// step: 32 MiB per step, the largest term of the step time).
//
// Grid barrier: 16 arrival counters on separate cache lines (16 arrivals each instead of 256 atomics serialised on
// one address), polled by the 16 low lanes of wave 0 with one load each; monotonic targets, bounded spin (on a
// timeout the error word is set and every workgroup runs to completion instead of hanging).
// Same arithmetic order per accumulator as lstm_wave_kernel (bit-identical results).
// -------------------------------------------------------------------------------------------------
struct LstmPersistArgs {
    const float *w0, *w1, *bias1, *xproj;
    float* hist;         // [BH zeros][T x BH: h0(0..T-1)][T x BH: h1(0..T-1)], BH = 16 * tiles * H; one slot = [batch tile][H/4][16 rows][4 units]:
                         // the 64 lanes of a consumer's B-fragment load read 1 KiB of CONTIGUOUS memory (lane (g, r) = 16 bytes at 16*lane),
                         // and the 64 producer lanes of a workgroup write one contiguous 256-byte piece.  (Row-major [B][H] put the 16 lanes
                         // of a fragment row 4 KiB apart -- every lane of a load on its own cache line: 7.2 us per step instead of 5.1.)
    float* y;
    unsigned* sync;      // 16 counters at [32*i], error flag at [512]; zeroed by the caller before every launch
    unsigned* status;    // host-visible engine status words (kernels.h FC_STATUS_*), or null
    int B, H, T;
    int ablate;          // FC_ABLATE_LSTM in FC_AB_KNOBS builds: 1 no grid barrier (profiling aid, wrong results).  Ignored (FC_ABL) in the shipped library
    int test_timeout;    // test hook (FC_ABLATE_LSTM=64, the only value the shipped library honours): behave as if the grid barrier had timed out
    int tile_base, tiles;   // first 16-row batch tile of this launch / tiles of the whole call (history slot size, barrier word block)
    int groups;          // independent recurrences in one launch: group j = workgroups [j*H/4, (j+1)*H/4) owns batch rows [16j, 16j+16) with
                         // its own barrier words (H = 512 fills only half of the chip: two batch tiles then advance side by side)
};

constexpr int kLstmSyncWords = 1024;
// profiling builds only (FC_BUILD_DEFINES="FC_LSTM_ABL=<mask>", results are garbage): 1 no critical-path MFMAs, 2 no hidden-state loads,
// 4 no shadow MFMAs, 8 no y stores, 16 no store drain before the arrival.  Compile time, because a run-time branch around the loads
// would make hipcc wait for them at the join.
#ifndef FC_LSTM_ABL
#define FC_LSTM_ABL 0
#endif

// split grid barrier: arrive (after this workgroup's stores have drained) ... independent work ... wait
__device__ __forceinline__ void lstm_barrier_arrive(unsigned* sync, int blk) {
    if (!(FC_LSTM_ABL & 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its own write-through stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(sync + 32 * (blk & 15), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lstm_barrier_wait(unsigned* sync, unsigned target) {
    if (threadIdx.x < 64) {
        const unsigned* mine = sync + 32 * (threadIdx.x & 15);
        unsigned spins = 0;
        while (true) {
            const unsigned v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v >= target)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0u) {
                if (__hip_atomic_load(sync + 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (spins > (1u << 21)) { __hip_atomic_store(sync + 512, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
    }
    __syncthreads();
}

// Wavefront schedule (T + 2 steps): at step s layer 0 produces h0(s) and layer 1 produces h1(s-2).  Layer 1 runs one
// step later than its data dependence requires so that its input half, P = W_ih1 . h0(s-1), is computed in the SHADOW
// of the grid barrier of step s (after this workgroup has arrived, before it starts polling) and only the two
// recurrent products (W_hh0 . h0(s-1), W_hh1 . h1(s-3)) sit on the critical path between two barriers.
template <int NS>
__global__ __launch_bounds__(256, 1) void lstm_persist_kernel(const LstmPersistArgs p) {
    // batch tiles (16 rows) per workgroup and step.  Fixed at one since round 3: a second tile of a call is a second group (H = 512) or a
    // second launch (H = 1024); the tile loops below are what is left of the two-tiles-per-step form (3.14 vs 2 x 1.26 ms) and fold away.
    constexpr int NBT = 1;
    static_assert(NS % 2 == 0, "k slices are issued in pairs");
    __shared__ f32x4 red[2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int H = p.H, B = p.B, T = p.T;
    const int wgs = H >> 2;                          // workgroups of one recurrence
    const int grp = p.groups > 1 ? blockIdx.x / wgs : 0;
    const int blk = blockIdx.x - grp * wgs;
    const int tile0 = p.tile_base + grp;             // first batch tile of this group (groups > 1 run with NBT == 1)
    const int brow0 = tile0 * 16;                    // its first batch row
    unsigned* const sync = p.sync + (size_t)tile0 * kLstmSyncWords;
    const unsigned arrivals = (unsigned)wgs >> 4;    // per counter per step
    const int kslice = NS * 16;                      // H / 4 waves
    const size_t BH = (size_t)16 * p.tiles * H;      // floats per history slot (batch padded to whole tiles)
    // ---- weights -> registers (once)
    f32x4 a0[NS], a1i[NS], a1h[NS];
    {
        const float* w0r = p.w0 + ((size_t)blk * 16 + r16) * H + wid * kslice + 4 * g;
        const float* w1r = p.w1 + ((size_t)blk * 16 + r16) * 2 * H + wid * kslice + 4 * g;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            a0[q] = *(const f32x4*)(w0r + 16 * q);
            a1i[q] = *(const f32x4*)(w1r + 16 * q);
            a1h[q] = *(const f32x4*)(w1r + H + 16 * q);
        }
    }
    float cst[NBT];                                  // cell state of (batch row, unit): wave 0 -> layer 0, wave 1 -> layer 1
    f32x4 pa[NBT], pb[NBT];                          // P = W_ih1 . h0 of the NEXT step's layer-1 timestep (even / odd k slices)
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) {
        cst[nb] = 0.f;
        pa[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pb[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 bias1 = *(const f32x4*)(p.bias1 + (size_t)blk * 16 + 4 * g);
    float* const hist0 = p.hist;                     // slot k = h0(k-1), slot 0 = zeros
    float* const hist1 = p.hist + (size_t)T * BH;    // slot k >= 1 = h1(k-1) (slot 0 is never addressed through this base)
    f32x4 b0keep[NS];

    for (int s = 0; s <= T + 1; ++s) {
        const bool act0 = s < T, act1 = s >= 2;
        const float* h0in = hist0 + (size_t)(s <= T ? s : T) * BH;                   // h0(s-1)
        const float* h1in = s >= 3 ? hist1 + (size_t)(s - 2) * BH : hist0;           // h1(s-3)
        float* h0o = hist0 + (size_t)(s + 1) * BH;                                   // h0(s)     (s < T)
        float* h1o = hist1 + (size_t)(s - 1) * BH;                                   // h1(s-2)   (s >= 2)
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb) {
            const int brow = brow0 + nb * 16 + r16;
            const bool bvalid = brow < B;
            // rows >= B of the last tile are never written: their columns compute values nobody stores (MFMA columns are independent)
            const size_t hoff = (size_t)(tile0 + nb) * 16 * H + (size_t)wid * kslice * 16 + 4 * lane;
            f32x4 xp = {0.f, 0.f, 0.f, 0.f};
            if (wid == 0 && act0)
                xp = *(const f32x4*)(p.xproj + ((size_t)s * B + (bvalid ? brow : 0)) * 4 * H + (size_t)blk * 16 + 4 * g);
            f32x4 b0l[NBT == 1 ? 1 : NS], b1[NS];
            f32x4 (&b0)[NS] = *(NBT == 1 ? &b0keep : (f32x4 (*)[NS])&b0l);   // single batch tile: h0(s-1) stays in registers for the shadow phase
            // All 2*NS loads are issued back to back in the order the MFMAs consume them and NOTHING touches the values before
            // their MFMA: the k slices then start as they arrive (counted vmcnt) instead of after the last one (round 1 masked
            // the columns of batch rows >= B right after the loads, which put one vmcnt(0) in front of all 128 MFMAs: 3.3 us
            // of loads and 2.1 us of MFMAs ran back to back).  Columns of rows >= B read row 0 and compute values nobody stores.
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                if (FC_LSTM_ABL & 2) { b0[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; b1[q] = b0[q]; continue; }
                b0[q] = *(const f32x4*)(h0in + hoff + 256 * q);
                b1[q] = *(const f32x4*)(h1in + hoff + 256 * q);
            }
            // four accumulators (layer 0 / layer 1 x even / odd k slice); the ISSUE order interleaves them so that two
            // MFMAs on the same accumulator are never back to back, the order WITHIN each accumulator is that of
            // lstm_wave_kernel (layer 1: the W_ih1 terms -- already in pa / pb -- then the W_hh1 terms)
            f32x4 c0a = {0.f, 0.f, 0.f, 0.f}, c0b = {0.f, 0.f, 0.f, 0.f};
            f32x4 c1a = pa[nb], c1b = pb[nb];
#pragma unroll
            for (int q = 0; q < ((FC_LSTM_ABL & 1) ? 0 : NS); q += 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q][j], b0[q][j], c0a, 0, 0, 0);
                    c1a = __builtin_amdgcn_mfma_f32_16x16x4f32(a1h[q][j], b1[q][j], c1a, 0, 0, 0);
                    c0b = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q + 1][j], b0[q + 1][j], c0b, 0, 0, 0);
                    c1b = __builtin_amdgcn_mfma_f32_16x16x4f32(a1h[q + 1][j], b1[q + 1][j], c1b, 0, 0, 0);
                }
            }
            red[0][wid][lane] = c0a + c0b;
            red[1][wid][lane] = c1a + c1b;
            __syncthreads();
            if (wid < 2) {
                const int layer = wid;
                const bool act = layer ? act1 : act0;
                f32x4 sg = red[layer][0][lane];
                for (int w = 1; w < 4; ++w) sg = sg + red[layer][w][lane];
                const f32x4 add = layer ? bias1 : xp;
                if (act && bvalid) {
                    const float gi = sigmoid_f(sg[0] + add[0]);
                    const float gf = sigmoid_f(sg[1] + add[1]);
                    const float gg = tanh_f(sg[2] + add[2]);
                    const float go = sigmoid_f(sg[3] + add[3]);
                    const float cn = lstm_cell(gf, cst[nb], gi, gg);
                    const float hn = go * tanh_f(cn);
                    cst[nb] = cn;
                    const size_t ci = (size_t)brow * H + (size_t)blk * 4 + g;
                    const size_t hi = (size_t)(tile0 + nb) * 16 * H + ((size_t)blk * 16 + r16) * 4 + g;
                    if (layer == 0) __hip_atomic_store(&h0o[hi], hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else { __hip_atomic_store(&h1o[hi], hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (!(FC_LSTM_ABL & 8)) p.y[ci * T + (s - 2)] = hn; }
                }
            }
            __syncthreads();
        }
        if (s > T) break;
        const bool do_sync = !FC_ABL(p.ablate, 1);
        if (do_sync) lstm_barrier_arrive(sync, blk);
        // ---- barrier shadow: P for the next step's layer-1 timestep t = s - 1 (needs h0(s-1), already visible)
        if (s >= 1) {
#pragma unroll
            for (int nb = 0; nb < NBT; ++nb) {
                const size_t hoff = (size_t)(tile0 + nb) * 16 * H + (size_t)wid * kslice * 16 + 4 * lane;
                f32x4 b0l[NBT == 1 ? 1 : NS];
                f32x4 (&b0)[NS] = *(NBT == 1 ? &b0keep : (f32x4 (*)[NS])&b0l);
                if (NBT > 1) {                        // several batch tiles: re-read (L2 hit) instead of holding NBT x 64 registers
#pragma unroll
                    for (int q = 0; q < NS; ++q) b0[q] = *(const f32x4*)(h0in + hoff + 256 * q);
                }
                f32x4 c1a = {0.f, 0.f, 0.f, 0.f}, c1b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < ((FC_LSTM_ABL & 4) ? 0 : NS); q += 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        c1a = __builtin_amdgcn_mfma_f32_16x16x4f32(a1i[q][j], b0[q][j], c1a, 0, 0, 0);
                        c1b = __builtin_amdgcn_mfma_f32_16x16x4f32(a1i[q + 1][j], b0[q + 1][j], c1b, 0, 0, 0);
                    }
                }
                pa[nb] = c1a; pb[nb] = c1b;
            }
        }
        if (do_sync) lstm_barrier_wait(sync, (unsigned)(s + 1) * arrivals);
    }
    // a barrier that timed out (some workgroup was not resident) must not pass for a result: poison this workgroup's
    // outputs so that the failure is loud downstream (the engine's per-step launch path is the supported fallback)
    if (__hip_atomic_load(sync + 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || p.test_timeout) {
        if (p.status && tid == 0) *(volatile unsigned*)(p.status + FC_STATUS_LSTM_TIMEOUT) = 1u;   // read by the host at its next fc_* call
        for (int i = tid; i < 4 * B * T; i += 256) {
            const int t = i % T, bu = i / T;
            p.y[((size_t)(bu / 4) * H + (size_t)blk * 4 + (bu & 3)) * T + t] = __builtin_nanf("");
        }
    }
}

// H = 512 occupies 128 of the 256 CUs: a second batch tile gets its own 128 workgroups (and barrier words) instead of a second pass
static int lstm_persist_groups(int B, int H) { return (H == 512 && B > 16) ? 2 : 1; }
// zero fill as a KERNEL of this library (not hipMemsetAsync): under HIP-graph replay the runtime's memset node was observed to run
// unordered with respect to the neighbouring kernel nodes (tools/graph_probe.py), which resets the barrier words mid-recurrence
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}
__global__ __launch_bounds__(256) void tanh_range_kernel(float* __restrict__ x, size_t n, float range) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = tanhf(x[i]) * range;
}
hipError_t launch_tanh_range(float* x, size_t n, float range, hipStream_t st) {
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(tanh_range_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, range);
    return hipGetLastError();
}
hipError_t launch_zero_fill(float* p, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n);
    return hipGetLastError();
}

// history slots hold whole 16-row batch tiles (LstmPersistArgs::hist)
static int lstm_persist_tiles(int B) { return (B + 15) / 16; }
static size_t lstm_persist_slot_floats(int B, int H) { return (size_t)16 * lstm_persist_tiles(B) * H; }
size_t lstm_persist_state_floats(int B, int H, int T) { return (size_t)kLstmSyncWords * lstm_persist_tiles(B) + (size_t)(2 * T + 1) * lstm_persist_slot_floats(B, H); }
size_t lstm_persist_clear_floats(int B, int H) { return (size_t)kLstmSyncWords * lstm_persist_tiles(B) + lstm_persist_slot_floats(B, H); }

// `state`: lstm_persist_state_floats() floats whose first lstm_persist_clear_floats() are zero (barrier words + the
// all-zero initial hidden state)
hipError_t launch_lstm_persist(const float* w0, const float* w1, const float* bias1, const float* xproj, float* state, float* y,
                               int B, int H, int T, unsigned* status, hipStream_t st) {
    LstmPersistArgs a;
    const int groups = lstm_persist_groups(B, H);
    const int tiles = lstm_persist_tiles(B);
    a.w0 = w0; a.w1 = w1; a.bias1 = bias1; a.xproj = xproj; a.hist = state + (size_t)kLstmSyncWords * tiles; a.y = y;
    a.sync = (unsigned*)state; a.status = status; a.B = B; a.H = H; a.T = T; a.groups = groups; a.tile_base = 0; a.tiles = tiles;
    // FC_ABLATE_LSTM: the profiling masks are tuning-build knobs (ab_knob); the shipped library honours exactly one value, 64 = the barrier-timeout
    // TEST hook of tests/test_gpu_parity.py (every other value is ignored: mask 1 removes the grid barrier and gives wrong results)
    static const int hook = getenv("FC_ABLATE_LSTM") ? atoi(getenv("FC_ABLATE_LSTM")) : 0;
#ifdef FC_AB_KNOBS
    static const int ablate = hook;
#else
    static const int ablate = 0;         // (not through ab_knob: the variable is legitimately set by the timeout test, no "ignored" notice)
#endif
    a.ablate = ablate & ~64;
    a.test_timeout = (hook == 64 || (ablate & 64)) ? 1 : 0;
    // H = 1024 fills the chip with ONE batch tile: a second tile (17 .. 32 utterances) is a second launch of the same kernel on its own
    // history columns and barrier words (round 3: 2 x 1.26 ms; the two-tiles-per-step instantiation needed 3.14 ms with the new history
    // layout).  H = 512: two tiles side by side as two groups of 128 workgroups in one launch.
    dim3 grid(H / 4 * groups), block(256);
    // FC_LSTM_COOP=1: hipLaunchCooperativeKernel (the runtime validates the grid against the occupancy query at every launch,
    // +15-19 us of host time per launch); default: plain launch, residency validated once per engine by lstm_persist_supported()
    static const int coop = ab_knob("FC_LSTM_COOP", 0);
    void* kargs[] = {(void*)&a};
#define FC_LP(NS)                                                                                                \
    do {                                                                                                         \
        if (coop) { hipError_t ec = hipLaunchCooperativeKernel((const void*)lstm_persist_kernel<NS>, grid, block, kargs, 0, st); \
                    if (ec != hipSuccess) return ec; }                                                            \
        else hipLaunchKernelGGL((lstm_persist_kernel<NS>), grid, block, 0, st, a);                                 \
    } while (0)
    if (H == 1024) {
        for (int tb = 0; tb < tiles; ++tb) { a.tile_base = tb; FC_LP(16); }
    } else if (H == 512 && groups == tiles) FC_LP(8);
    else return hipErrorInvalidValue;
#undef FC_LP
    return hipGetLastError();
}

// can the persistent kernel be used? (every workgroup must be co-resident: one per CU)
// The grid barrier needs all H/4 workgroups resident at once.  Checked against the runtime's own occupancy answer for the
// instantiation that would run (not just the CU count): blocks per CU x CUs >= grid, with the one-block margin the
// microarchitecture guide asks for near an SGPR edge (the kernel needs exactly ONE block per CU on a 256-CU part, and
// two fit by registers, so the margin is met by construction).
bool lstm_persist_supported(int B, int H, int L, int device) {
    if (L != 2 || (H != 1024 && H != 512) || B > 32) return false;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
    const int groups = lstm_persist_groups(B, H);
    int per_cu = 0;
    hipError_t e = hipErrorInvalidValue;
    if (H == 1024) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_persist_kernel<16>, 256, 0);
    else if (H == 512) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_persist_kernel<8>, 256, 0);
    if (e != hipSuccess || per_cu < 1) return false;
    return (long long)per_cu * prop.multiProcessorCount >= H / 4 * groups && prop.multiProcessorCount >= H / 4 * groups;
}

// =================================================================================================
// 6. Triangle-weighted overlap-add of decoded segments (_linear_overlap_add, codec_basic.py:77-116)
//    weight = 0.5 - |t - 0.5|, t = linspace(0, 1, L0 + 2)[1:-1] with L0 the FIRST frame's length; out = sum_f w*frame_f
//    accumulated in frame order (product rounded, then added: the reference's `out += weight * frame`), divided once by
//    the summed weights.  Frame f starts at f*stride and is lens[f] samples long ([B][lens[f]] contiguous).
// =================================================================================================
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* const* __restrict__ frames, const int* __restrict__ lens,
                                                          int n_frames, int L0, int stride, int out_len, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int steps = L0 + 2, halfway = steps / 2;
    const float step = 1.0f / (float)(steps - 1);
    for (int pos = blockIdx.x * 256 + threadIdx.x; pos < out_len; pos += gridDim.x * 256) {
        int f_lo = pos - L0 + 1;
        f_lo = f_lo <= 0 ? 0 : (f_lo + stride - 1) / stride;
        int f_hi = pos / stride;
        if (f_hi > n_frames - 1) f_hi = n_frames - 1;
        float acc = 0.f, wsum = 0.f;
        for (int f = f_lo; f <= f_hi; ++f) {
            const int i = pos - f * stride, n = lens[f];
            if (i >= n) continue;
            const int idx = i + 1;                                     // linspace(...)[1:-1]
            const float t = idx < halfway ? step * (float)idx : 1.0f - step * (float)(steps - idx - 1);
            const float w = 0.5f - fabsf(t - 0.5f);
            acc = __fadd_rn(acc, __fmul_rn(w, frames[f][(size_t)b * n + i]));
            wsum = __fadd_rn(wsum, w);
        }
        out[(size_t)b * out_len + pos] = acc / wsum;
    }
}

hipError_t launch_overlap_add(const float* const* frames, const int* lens, int n_frames, int B, int L0, int stride, int out_len,
                              float* out, hipStream_t st) {
    if (n_frames <= 0 || B <= 0 || out_len <= 0) return hipSuccess;
    int gx = ceil_div(out_len, 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(overlap_add_kernel, dim3(gx, B), dim3(256), 0, st, frames, lens, n_frames, L0, stride, out_len, out);
    return hipGetLastError();
}

}  // namespace fc
