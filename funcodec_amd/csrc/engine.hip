// Engine: layer plan, checkpoint ingestion / weight re-layout, workspace planning and the C ABI
// (include/funcodec_amd.h).  All device work is enqueued on the caller's stream; nothing here
// synchronises with the host after fc_engine_finalize().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/funcodec_amd.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;

int fail(const std::string& msg) { g_err = msg; return 1; }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

struct HostTensor {
    std::vector<int64_t> dims;
    std::vector<float> data;
    bool set = false;
};

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
};

// One SConv1d / SConvTranspose1d (+ its GroupNorm) of the plan.
struct ConvLayer {
    std::string prefix;     // e.g. "encoder.model.3.conv" / "decoder.model.3.convtr"
    bool transposed = false;
    int cin = 0, cout = 0, k = 1, stride = 1, dil = 1;
    bool has_norm = true;   // GroupNorm(1, cout) after the conv
    bool wnorm = false;     // checkpoint stores weight_g / weight_v (torch.nn.utils.weight_norm)
    bool causal = false;
    bool dual = false;      // the plan feeds it two summed sources (residual sum / LSTM skip)
    bool small_n = false;   // runs at the bottleneck frame rate: few columns per utterance
    // 2-D layers (STFT-domain codec), frequency-major layout: a Conv2d (kf x k, stride sf x stride) over c2d channels is the 1-D
    // conv (k, stride) over cin = kf * c2d channels of the virtual utterance starting at frequency row fo * sf
    int kf = 1, sf = 1, c2d = 0;
    bool extra_left = false;    // SConv2d puts the "extra" time padding on the LEFT (conv.py:377), SConv1d on the right
    bool valid = false;         // no padding at all (the STFT GEMM)
    int zpadL = -1, zpadR = -1; // explicit ZERO padding (the inverse-STFT GEMM); -1: reference padding rules
    bool synthetic = false;     // weights generated at finalize (DFT matrices), not part of the checkpoint
    int groups = 1;             // grouped Conv2d / ConvTranspose2d (conv_group_ratio > 0): dense block-diagonal weight at finalize
    float* w_group = nullptr;   // grouped Conv2d with 2 / 4 channels per group: torch-layout weights for the direct kernel (freq_kernels.hip)
    bool force_plain = false;   // the slab of the fused-prologue variants does not fit (large strides): materialise the input first
    bool unsupported = false;   // no variant fits (stride > ~17): refused at fc_engine_create
    // GEMM view
    int M = 0, gk = 1, gstride = 1;    // rows, taps, stride of the implicit GEMM
    int BM = 128, BN = 128, CC = 2, nchunk = 1, Mpad = 0;
    bool row = false;       // stride-1 row staging (kernels.hip)
    float* w_plain = nullptr;   // device [cin][k] when cout == 1 (FMA kernel instead of a 32-row MFMA tile)
    float bias0 = 0.f;
    float *wt = nullptr, *bias = nullptr, *gamma = nullptr, *beta = nullptr;   // device
    int* koff = nullptr;                                                        // device
};

struct LstmLayer {
    ConvLayer inproj;       // W_ih as a k=1 GEMM with permuted rows, bias = b_ih + b_hh
    float* whh = nullptr;   // [4H][H] rows permuted to (blk, unit, gate)
    float* wcat = nullptr;  // layers >= 1: [4H][2H] = [W_ih | W_hh], same row permutation (wavefront kernel)
    float* bperm = nullptr; // layers >= 1: permuted b_ih + b_hh
};

struct LstmBlock {
    std::string prefix;     // "encoder.model.16.lstm"
    int H = 0;
    std::vector<LstmLayer> layers;
};

struct Act {               // raw tensor [B][C][T] + pending GroupNorm affine (null = already final)
    float* raw = nullptr;
    float* aff = nullptr;
    int C = 0, T = 0;
    bool normed = false;   // has a pending affine (valid in dry-run planning too, where pointers are null)
};

struct Ctx {
    int B = 0;
    hipStream_t st = nullptr;
    char* base = nullptr;
    size_t cap = 0, off = 256;       // the first buffer starts 256 bytes in: the grouped 3 x 3 conv reads one float in front of a row (freq_kernels.hip, FASTEDGE)
    bool dry = false;
    int err = 0;
    static constexpr size_t kTailSlack = 4096;
    int launches = 0, conv_launches = 0;
    double conv_flops = 0, conv_bytes = 0, lstm_flops = 0, rvq_flops = 0, other_bytes = 0;
    template <typename T>
    T* alloc(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = dry ? nullptr : (T*)(base + off);
        off += n * sizeof(T);
        // every buffer ends at least kTailSlack bytes before the end of the workspace: kernels with unclamped row-end loads (the grouped
        // 3 x 3 / 8 x 2 convs, freq_kernels.hip FASTEDGE) and the DMA-staged conv (over-read of the last N tile) rely on it
        if (!dry && off + kTailSlack > cap) { err = 1; g_err = "workspace too small (sizes from fc_engine_workspace_bytes include 4 KiB of tail slack)"; return nullptr; }
        return p;
    }
};

int ceil_div_i(int a, int b) { return (a + b - 1) / b; }

const char* kLstmPersistClass = "lstm_persist_kernel<NS> (whole recurrence of one SLSTM block, one launch)";
const char* kLstmWaveClass = "lstm_wave_kernel<NS> (whole recurrence of one SLSTM block: T + L - 1 launches incl. gaps)";
const char* kRvqClass = "rvq_encode_kernel<D, RS> (all stages of the residual quantiser)";

}  // namespace

struct fc_engine {
    fc_arch arch;
    int device = 0;
    bool finalized = false;
    std::vector<std::pair<std::string, std::vector<int64_t>>> expected;   // checkpoint contract, in plan order
    std::map<std::string, HostTensor> host;
    // plan
    ConvLayer enc_first, enc_last, dec_first, dec_last;
    struct ResBlock {                                                     // SEANetResnetBlock: shortcut(x) + block(x)
        ConvLayer shortcut, block1, block3;
        bool fused_head = false;                // shortcut + block.1 in one launch (thin blocks, kernels.hip 1c)
        float *wsc = nullptr, *wb1 = nullptr;   // LDS images of the two weight matrices for that kernel
        float *bsc = nullptr, *bb1 = nullptr;
    };
    struct Stage { std::vector<ResBlock> res; ConvLayer resample; };      // n_residual_layers blocks; resample = down (enc) / up (dec)
    std::vector<Stage> enc_stages, dec_stages;
    LstmBlock enc_lstm, dec_lstm;
    std::map<std::string, ConvLayer*> by_prefix;
    std::map<std::string, ResBlock*> res_by_prefix;
    // STFT-domain codec (arch.model_type == 1)
    ConvLayer enc2_first, dec2_last, stft, istft;
    ConvLayer q_in, q_out;                                                 // CostumeQuantizer.input_proj / output_proj as k = 1 GEMMs (codec_dim != dimension)
    bool q_proj = false;
    int cdim() const { return q_proj ? arch.codec_dim : arch.dimension; }
    // audio channels of the time-domain codec (wav / recon buffers are [B][C][T]); FreqCodec is mono (codec_freq.py squeezes dim 1)
    int audio_ch() const { return arch.model_type == 0 && arch.input_channels == 2 ? 2 : 1; }
    std::vector<std::vector<ConvLayer>> dec_up_phases;                    // decoder stage s: one transposed 1-D GEMM per frequency phase
    int halo2 = 3;                                                        // frequency halo rows of the 2-D activations
    float* win2 = nullptr;                                                // squared Hann window [n_fft] (istft envelope)                       // "encoder.model.1" -> block (fc_resblock_forward)
    std::map<std::string, LstmBlock*> lstm_by_prefix;
    // quantiser
    float *cb = nullptr, *enorm = nullptr;   // [nq][K][D], [nq][K]
    float* cb_frag = nullptr;                // the codebooks in MFMA B-fragment order (kernels.hip rvq_encode_kernel), or null
    std::vector<void*> dev_allocs;
    // status words written by kernels (host-pinned, device-mapped; kernels.h FC_STATUS_*), read at the next call without a sync
    volatile unsigned* status_host = nullptr;
    unsigned* status_dev = nullptr;
    bool lstm_persist_ok = true;             // cleared after a grid-barrier timeout: later calls take the per-step launch path
    int persist_checked_B = -1; bool persist_checked_val = false;
    // optional event timing
    bool profiling = false;
    struct Span { hipEvent_t a, b; int cls; double flops, bytes; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    std::vector<std::string> prof_names;     // class id -> kernel name as rocprofv3 prints it
    int prof_class(const std::string& name) {
        for (size_t i = 0; i < prof_names.size(); ++i)
            if (prof_names[i] == name) return (int)i;
        if ((int)prof_names.size() >= FC_PROF_CLASSES) return FC_PROF_CLASSES - 1;
        prof_names.push_back(name);
        return (int)prof_names.size() - 1;
    }
};

namespace {

hipEvent_t prof_event(fc_engine* e) {
    if (e->events_used == e->event_pool.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        e->event_pool.push_back(ev);
    }
    return e->event_pool[e->events_used++];
}

struct ProfSpan {   // RAII: records start now, stop at scope exit
    fc_engine* e; hipStream_t st; int idx = -1;
    ProfSpan(fc_engine* e_, Ctx& cx, int cls, double flops, double bytes) : e(e_), st(cx.st) {
        if (!e->profiling || cx.dry || cx.err) return;
        fc_engine::Span s; s.a = prof_event(e); s.b = prof_event(e); s.cls = cls; s.flops = flops; s.bytes = bytes;
        if (!s.a || !s.b) return;
        (void)hipEventRecord(s.a, st);
        e->spans.push_back(s);
        idx = (int)e->spans.size() - 1;
    }
    ~ProfSpan() { if (idx >= 0) (void)hipEventRecord(e->spans[idx].b, st); }
};

// ---- plan construction (mirrors nn.Sequential indices: seanet_encoder.py:109-160, seanet_decoder.py:111-164)
void add_conv_expect(fc_engine* e, ConvLayer& L) {
    const std::string inner = L.transposed ? ".convtr" : ".conv";
    const int d0 = L.transposed ? L.cin : L.cout, d1 = L.transposed ? L.cout : L.cin;
    if (L.wnorm) {
        e->expected.push_back({L.prefix + inner + ".weight_g", {d0, 1, 1}});
        e->expected.push_back({L.prefix + inner + ".weight_v", {d0, d1, L.k}});
    } else {
        e->expected.push_back({L.prefix + inner + ".weight", {d0, d1, L.k}});
    }
    e->expected.push_back({L.prefix + inner + ".bias", {L.cout}});
    if (L.has_norm) {
        e->expected.push_back({L.prefix + ".norm.weight", {L.cout}});
        e->expected.push_back({L.prefix + ".norm.bias", {L.cout}});
    }
    e->by_prefix[L.prefix] = &L;
}

void choose_tiling(ConvLayer& L);

ConvLayer mk_conv(const std::string& prefix, int cin, int cout, int k, int stride, bool transposed = false,
                  bool dual = false, bool small_n = false, int dil = 1);
ConvLayer mk_conv_(const std::string& prefix, int cin, int cout, int k, int stride, bool transposed, bool dual, bool small_n, int dil);
ConvLayer mk_conv(const std::string& prefix, int cin, int cout, int k, int stride, bool transposed, bool dual, bool small_n, int dil) {
    ConvLayer L = mk_conv_(prefix, cin, cout, k, stride, transposed, dual, small_n, dil);
    if (getenv("FC_DUMP_PLAN"))     // tuning aid: the tiling every layer gets
        fprintf(stderr, "plan %-36s cin=%4d M=%4d k=%2d s=%d  tile %3dx%3d CC=%2d nchunk=%3d %s\n", L.prefix.c_str(), L.cin, L.M,
                L.gk, L.gstride, L.BM, L.BN, L.CC, L.nchunk, L.row ? "row" : "");
    return L;
}
ConvLayer mk_conv_(const std::string& prefix, int cin, int cout, int k, int stride, bool transposed, bool dual, bool small_n, int dil) {
    ConvLayer L;
    L.prefix = prefix; L.cin = cin; L.cout = cout; L.k = k; L.stride = stride; L.transposed = transposed; L.dil = dil;
    L.dual = dual; L.small_n = small_n;
    if (!transposed) { L.M = cout; L.gk = k; L.gstride = stride; }
    else { L.M = cout * stride; L.gk = 2; L.gstride = 1; }   // 2-tap GEMM over the r output phases
    choose_tiling(L);
    return L;
}

void choose_tiling(ConvLayer& L);

// ---- STFT-domain codec: SEANetEncoder2d / SEANetDecoder2d (seanet_encoder.py:252-363, seanet_decoder.py:244-360) ---------------
// Same Sequential indexing as the reference; every Conv2d is planned as a 1-D GEMM over kf * C channels (frequency-major layout).
// `groups` > 1 (conv_group_ratio > 0): the layer still runs as the dense GEMM over a block-diagonal weight built at finalize (the
// zero products add exactly nothing: same result as the grouped conv accumulated in the same order)
ConvLayer mk_conv2d(const std::string& prefix, int cin, int cout, int kf, int kt, int sf, int st, bool dual, int groups = 1) {
    ConvLayer L = mk_conv(prefix, kf * cin, cout, kt, st, false, dual, false, 1);
    L.kf = kf; L.sf = sf; L.c2d = cin; L.extra_left = true; L.groups = groups;
    return L;
}
inline int group_count(int n, int ratio) { return ratio > 0 ? n / 2 / ratio : 1; }

void add_conv2d_expect(fc_engine* e, ConvLayer& L) {
    const int cpg = L.c2d / (L.groups > 0 ? L.groups : 1);                         // groups < 1 is refused at create
    if (L.wnorm) {                                                                 // torch.nn.utils.weight_norm, dim 0 (conv.py:24-25)
        e->expected.push_back({L.prefix + ".conv.weight_g", {L.cout, 1, 1, 1}});
        e->expected.push_back({L.prefix + ".conv.weight_v", {L.cout, cpg, L.kf, L.k}});
    } else {
        e->expected.push_back({L.prefix + ".conv.weight", {L.cout, cpg, L.kf, L.k}});
    }
    e->expected.push_back({L.prefix + ".conv.bias", {L.cout}});
    if (L.has_norm) {
        e->expected.push_back({L.prefix + ".norm.weight", {L.cout}});
        e->expected.push_back({L.prefix + ".norm.bias", {L.cout}});
    }
    e->by_prefix[L.prefix] = &L;
}

// the weight a weight-normed 2-D layer computes in its forward pre-hook: v * (g / ||v||), the norm over every dim but 0 (`key` = prefix +
// ".conv" / ".convtr"; d0 = out channels of Conv2d, IN channels of ConvTranspose2d); the plain weight otherwise
std::vector<float> folded_weight_nd(fc_engine* e, const std::string& key, bool wnorm, size_t d0) {
    if (!wnorm) return e->host[key + ".weight"].data;
    const auto& V = e->host[key + ".weight_v"].data;
    const auto& G = e->host[key + ".weight_g"].data;
    const size_t inner_n = V.size() / d0;
    std::vector<float> folded(V.size());
    for (size_t r = 0; r < d0; ++r) {
        double ss = 0.0;
        for (size_t j = 0; j < inner_n; ++j) ss += (double)V[r * inner_n + j] * (double)V[r * inner_n + j];
        const float sc = G[r] / (float)sqrt(ss);
        for (size_t j = 0; j < inner_n; ++j) folded[r * inner_n + j] = V[r * inner_n + j] * sc;
    }
    return folded;
}

// CostumeQuantizer (costume_quantizer.py:7-55): the codebooks, and with codec_dim != input_size the two Linears around them
void add_quantizer_expect(fc_engine* e) {
    const fc_arch& a = e->arch;
    e->q_proj = a.codec_dim > 0 && a.codec_dim != a.dimension;
    if (e->q_proj) {
        e->q_in = mk_conv("quantizer.input_proj", a.dimension, a.codec_dim, 1, 1, false, false, true);
        e->q_out = mk_conv("quantizer.output_proj", a.codec_dim, a.dimension, 1, 1, false, false, true);
        e->q_in.has_norm = e->q_out.has_norm = false;
        e->expected.push_back({"quantizer.input_proj.weight", {a.codec_dim, a.dimension}});
        e->expected.push_back({"quantizer.input_proj.bias", {a.codec_dim}});
        e->expected.push_back({"quantizer.output_proj.weight", {a.dimension, a.codec_dim}});
        e->expected.push_back({"quantizer.output_proj.bias", {a.dimension}});
    }
    e->expected.push_back({"quantizer.rq.model.embed", {a.num_quantizers, a.codebook_size, e->cdim()}});
}

void build_plan_2d(fc_engine* e) {
    const fc_arch& a = e->arch;
    const int nf = a.n_filters, nres = a.n_residual_layers;
    e->halo2 = std::max(std::max(a.kernel_size, a.last_kernel_size), a.residual_kernel_size) / 2;
    for (int s = 0; s < a.n_ratios; ++s) e->halo2 = std::max(e->halo2, (a.ratios_f[s] + 1) / 2);
    auto name = [](const char* side, int idx, const char* suffix) { return std::string(side) + ".model." + std::to_string(idx) + suffix; };
    auto add_res = [&](fc_engine::ResBlock& R, const char* side, int idx, int c, int j, int dil) {
        const int hid = c / a.compress, rk = a.residual_kernel_size;
        const int gr = side[0] == 'e' ? a.enc_conv_group_ratio : a.dec_conv_group_ratio;
        R.shortcut = mk_conv2d(name(side, idx, ".shortcut.conv"), c, c, 1, 1, 1, 1, j > 0, group_count(c, gr));
        R.block1 = mk_conv2d(name(side, idx, ".block.1.conv"), c, hid, rk, rk, 1, 1, j > 0, group_count(hid, gr));
        R.block1.dil = dil;                                   // dilation (1, dilation_base ** j): time axis only
        R.block3 = mk_conv2d(name(side, idx, ".block.3.conv"), hid, c, 1, 1, 1, 1, false, group_count(hid, gr));
    };
    // ---- encoder
    int idx = 0, mult = 1;
    e->enc2_first = mk_conv2d(name("encoder", idx, ".conv"), a.input_channels, nf, a.kernel_size, a.kernel_size, 1, 1, false);
    idx++;
    e->enc_stages.resize(a.n_ratios);
    for (int s = 0; s < a.n_ratios; ++s) {
        const int fr = a.ratios_f[a.n_ratios - 1 - s], tr = a.ratios[a.n_ratios - 1 - s];
        const int c = mult * nf;
        auto& S = e->enc_stages[s];
        S.res.resize(nres);
        for (int j = 0, dil = 1; j < nres; ++j, dil *= a.dilation_base) { add_res(S.res[j], "encoder", idx, c, j, dil); idx++; }
        idx++;                                                                    // ELU
        S.resample = mk_conv2d(name("encoder", idx, ".conv"), c, 2 * c, 2 * fr, 2 * tr, fr, tr, true, group_count(c, a.enc_conv_group_ratio));
        idx++;
        mult *= 2;
    }
    idx++;                                                                        // ReshapeModule (squeeze the frequency axis)
    const int cb = mult * nf;
    if (a.lstm_layers > 0) { e->enc_lstm.prefix = name("encoder", idx, ".lstm"); e->enc_lstm.H = cb; idx++; }
    idx++;
    const bool skip_dual = a.lstm_layers > 0 && a.lstm_skip;
    e->enc_last = mk_conv(name("encoder", idx, ".conv"), cb, a.dimension, a.last_kernel_size, 1, false, skip_dual, true);
    // ---- decoder
    idx = 0;
    e->dec_first = mk_conv(name("decoder", idx, ".conv"), a.dimension, cb, a.kernel_size, 1, false, false, true);
    idx++;
    if (a.lstm_layers > 0) { e->dec_lstm.prefix = name("decoder", idx, ".lstm"); e->dec_lstm.H = cb; idx++; }
    idx++;                                                                        // ReshapeModule (unsqueeze)
    e->dec_stages.resize(a.n_ratios);
    e->dec_up_phases.resize(a.n_ratios);
    for (int s = 0; s < a.n_ratios; ++s) {
        const int fr = a.ratios_f[s], tr = a.ratios[s];
        const int c = mult * nf, c2 = c / 2;
        auto& S = e->dec_stages[s];
        idx++;                                                                    // ELU
        // ConvTranspose2d(c -> c2, (2 fr, 2 tr), stride (fr, tr)): S.resample only carries the checkpoint contract; the launches
        // are one transposed 1-D GEMM (2 c channels = frequency rows fi - 1, fi) per output-frequency phase
        S.resample = ConvLayer();
        S.resample.prefix = name("decoder", idx, ".convtr");
        S.resample.transposed = true;
        S.resample.cin = c; S.resample.cout = c2; S.resample.c2d = c; S.resample.kf = 2 * fr; S.resample.sf = fr;
        S.resample.k = 2 * tr; S.resample.stride = tr; S.resample.groups = group_count(c, a.dec_tr_conv_group_ratio);
        for (int p = 0; p < fr; ++p)
            e->dec_up_phases[s].push_back(mk_conv(S.resample.prefix + ".phase" + std::to_string(p), 2 * c, c2, 2 * tr, tr, true, false, false));
        idx++;
        S.res.resize(nres);
        for (int j = 0, dil = 1; j < nres; ++j, dil *= a.dilation_base) { add_res(S.res[j], "decoder", idx, c2, j, dil); idx++; }
        mult /= 2;
    }
    idx++;
    e->dec2_last = mk_conv2d(name("decoder", idx, ".conv"), nf, a.input_channels, a.last_kernel_size, a.last_kernel_size, 1, 1, true);
    // ---- the STFT and its inverse as GEMMs over the hop-phase view of the signal (weights generated at finalize)
    const int F = a.n_fft / 2 + 1, taps = ceil_div_i(a.n_fft, a.stft_hop);
    e->stft = mk_conv("stft", a.stft_hop, 2 * F, taps, 1);
    e->stft.valid = true; e->stft.has_norm = false; e->stft.synthetic = true;
    {   // norm / causality of every layer of the nets (norm_type 0: GroupNorm(1, C) after each conv; 1: weight_norm, optionally causal in time)
        auto flag = [&](ConvLayer& L) { L.has_norm = a.norm_type == 0; L.wnorm = a.norm_type == 1; L.causal = a.causal != 0; };
        flag(e->enc2_first); flag(e->enc_last); flag(e->dec_first); flag(e->dec2_last);
        for (auto* stages : {&e->enc_stages, &e->dec_stages})
            for (auto& S : *stages) {
                flag(S.resample);
                for (auto& R : S.res) { flag(R.shortcut); flag(R.block1); flag(R.block3); }
            }
        for (auto& ph : e->dec_up_phases)
            for (auto& L : ph) flag(L);
    }
    e->istft = mk_conv("istft", 2 * F, a.stft_hop, taps, 1);
    e->istft.zpadL = taps - 1; e->istft.zpadR = taps - 1; e->istft.has_norm = false; e->istft.synthetic = true;
    // ---- checkpoint contract, in execution order
    add_conv2d_expect(e, e->enc2_first);
    for (auto& S : e->enc_stages) {
        for (auto& R : S.res) { add_conv2d_expect(e, R.block1); add_conv2d_expect(e, R.block3); add_conv2d_expect(e, R.shortcut); }
        add_conv2d_expect(e, S.resample);
    }
    auto add_lstm = [&](LstmBlock& lb) {
        if (lb.H == 0) return;
        lb.layers.resize(a.lstm_layers);
        for (int l = 0; l < a.lstm_layers; ++l) {
            const std::string sfx = "_l" + std::to_string(l);
            lb.layers[l].inproj = mk_conv(lb.prefix + ".inproj" + sfx, lb.H, 4 * lb.H, 1, 1, false, false, true);
            lb.layers[l].inproj.has_norm = false;
            e->expected.push_back({lb.prefix + ".weight_ih" + sfx, {4 * lb.H, lb.H}});
            e->expected.push_back({lb.prefix + ".weight_hh" + sfx, {4 * lb.H, lb.H}});
            e->expected.push_back({lb.prefix + ".bias_ih" + sfx, {4 * lb.H}});
            e->expected.push_back({lb.prefix + ".bias_hh" + sfx, {4 * lb.H}});
        }
        e->lstm_by_prefix[lb.prefix] = &lb;
    };
    add_lstm(e->enc_lstm);
    add_conv_expect(e, e->enc_last);
    add_conv_expect(e, e->dec_first);
    add_lstm(e->dec_lstm);
    for (auto& S : e->dec_stages) {
        const int opg_t = S.resample.cout / (S.resample.groups > 0 ? S.resample.groups : 1);
        if (S.resample.wnorm) {                                               // dim 0 of a ConvTranspose2d weight = its IN channels
            e->expected.push_back({S.resample.prefix + ".convtr.weight_g", {S.resample.c2d, 1, 1, 1}});
            e->expected.push_back({S.resample.prefix + ".convtr.weight_v", {S.resample.c2d, opg_t, S.resample.kf, S.resample.k}});
        } else {
            e->expected.push_back({S.resample.prefix + ".convtr.weight", {S.resample.c2d, opg_t, S.resample.kf, S.resample.k}});
        }
        e->expected.push_back({S.resample.prefix + ".convtr.bias", {S.resample.cout}});
        if (S.resample.has_norm) {
            e->expected.push_back({S.resample.prefix + ".norm.weight", {S.resample.cout}});
            e->expected.push_back({S.resample.prefix + ".norm.bias", {S.resample.cout}});
        }
        for (auto& R : S.res) { add_conv2d_expect(e, R.block1); add_conv2d_expect(e, R.block3); add_conv2d_expect(e, R.shortcut); }
    }
    add_conv2d_expect(e, e->dec2_last);
    add_quantizer_expect(e);
}

void build_plan(fc_engine* e) {
    if (e->arch.model_type == 1) { build_plan_2d(e); return; }
    const fc_arch& a = e->arch;
    const int nf = a.n_filters, nres = a.n_residual_layers;
    auto name = [](const char* side, int idx, const char* suffix) {
        return std::string(side) + ".model." + std::to_string(idx) + suffix;
    };
    // ---- encoder
    int idx = 0, mult = 1;
    e->enc_first = mk_conv(name("encoder", idx, ".conv"), e->audio_ch(), nf, a.kernel_size, 1);
    idx++;
    e->enc_stages.resize(a.n_ratios);
    for (int s = 0; s < a.n_ratios; ++s) {
        const int ratio = a.ratios[a.n_ratios - 1 - s];
        const int c = mult * nf, hid = c / a.compress;
        auto& S = e->enc_stages[s];
        S.res.resize(nres);
        for (int j = 0, dil = 1; j < nres; ++j, dil *= a.dilation_base) {      // dilations [dilation_base**j, 1] (seanet_encoder.py:127-133)
            auto& R = S.res[j];
            R.shortcut = mk_conv(name("encoder", idx, ".shortcut.conv"), c, c, 1, 1, false, j > 0);
            R.block1 = mk_conv(name("encoder", idx, ".block.1.conv"), c, hid, a.residual_kernel_size, 1, false, j > 0, false, dil);
            R.block3 = mk_conv(name("encoder", idx, ".block.3.conv"), hid, c, 1, 1);
            idx++;
        }
        idx++;                                                                    // ELU
        S.resample = mk_conv(name("encoder", idx, ".conv"), c, 2 * c, 2 * ratio, ratio, false, true, s == a.n_ratios - 1);
        idx++;
        mult *= 2;
    }
    const int cb = mult * nf;
    if (a.lstm_layers > 0) {
        e->enc_lstm.prefix = name("encoder", idx, ".lstm");
        e->enc_lstm.H = cb;
        idx++;
    }
    idx++;
    const bool skip_dual = a.lstm_layers > 0 && a.lstm_skip;
    e->enc_last = mk_conv(name("encoder", idx, ".conv"), cb, a.dimension, a.last_kernel_size, 1, false, skip_dual, true);
    // ---- decoder
    idx = 0;
    e->dec_first = mk_conv(name("decoder", idx, ".conv"), a.dimension, cb, a.kernel_size, 1, false, false, true);
    idx++;
    if (a.lstm_layers > 0) {
        e->dec_lstm.prefix = name("decoder", idx, ".lstm");
        e->dec_lstm.H = cb;
        idx++;
    }
    e->dec_stages.resize(a.n_ratios);
    mult = 1 << a.n_ratios;
    for (int s = 0; s < a.n_ratios; ++s) {
        const int ratio = a.ratios[s];
        const int c = mult * nf, c2 = c / 2, hid = c2 / a.compress;
        auto& S = e->dec_stages[s];
        idx++;
        S.resample = mk_conv(name("decoder", idx, ".convtr"), c, c2, 2 * ratio, ratio, true, s > 0 || skip_dual, s == 0);
        idx++;
        S.res.resize(nres);
        for (int j = 0, dil = 1; j < nres; ++j, dil *= a.dilation_base) {
            auto& R = S.res[j];
            R.shortcut = mk_conv(name("decoder", idx, ".shortcut.conv"), c2, c2, 1, 1, false, j > 0);
            R.block1 = mk_conv(name("decoder", idx, ".block.1.conv"), c2, hid, a.residual_kernel_size, 1, false, j > 0, false, dil);
            R.block3 = mk_conv(name("decoder", idx, ".block.3.conv"), hid, c2, 1, 1);
            idx++;
        }
        mult /= 2;
    }
    idx++;
    e->dec_last = mk_conv(name("decoder", idx, ".conv"), nf, e->audio_ch(), a.last_kernel_size, 1, false, true);

    // ---- conv wrapper flavour of every SConv1d / SConvTranspose1d of the nets (conv.py:20-56)
    {
        std::vector<ConvLayer*> all = {&e->enc_first, &e->enc_last, &e->dec_first, &e->dec_last};
        for (auto* st : {&e->enc_stages, &e->dec_stages})
            for (auto& S : *st) {
                for (auto& R : S.res) { all.push_back(&R.shortcut); all.push_back(&R.block1); all.push_back(&R.block3); }
                all.push_back(&S.resample);
            }
        for (ConvLayer* L : all) {
            L->has_norm = a.norm_type == 0;
            L->wnorm = a.norm_type == 1;
            L->causal = a.causal != 0;
        }
    }
    for (auto* stg : {&e->enc_stages, &e->dec_stages})
        for (auto& S : *stg)
            for (auto& R : S.res) {
                const std::string& sp = R.shortcut.prefix;               // "<side>.model.<i>.shortcut.conv"
                e->res_by_prefix[sp.substr(0, sp.size() - std::string(".shortcut.conv").size())] = &R;
            }
    // ---- checkpoint contract, in execution order
    add_conv_expect(e, e->enc_first);
    for (auto& S : e->enc_stages) {
        for (auto& R : S.res) { add_conv_expect(e, R.shortcut); add_conv_expect(e, R.block1); add_conv_expect(e, R.block3); }
        add_conv_expect(e, S.resample);
    }
    auto add_lstm = [&](LstmBlock& lb) {
        if (lb.H == 0) return;
        lb.layers.resize(a.lstm_layers);
        for (int l = 0; l < a.lstm_layers; ++l) {
            const std::string sfx = "_l" + std::to_string(l);
            lb.layers[l].inproj = mk_conv(lb.prefix + ".inproj" + sfx, lb.H, 4 * lb.H, 1, 1, false, false, true);
            lb.layers[l].inproj.has_norm = false;
            e->expected.push_back({lb.prefix + ".weight_ih" + sfx, {4 * lb.H, lb.H}});
            e->expected.push_back({lb.prefix + ".weight_hh" + sfx, {4 * lb.H, lb.H}});
            e->expected.push_back({lb.prefix + ".bias_ih" + sfx, {4 * lb.H}});
            e->expected.push_back({lb.prefix + ".bias_hh" + sfx, {4 * lb.H}});
        }
        e->lstm_by_prefix[lb.prefix] = &lb;
    };
    add_lstm(e->enc_lstm);
    add_conv_expect(e, e->enc_last);
    add_conv_expect(e, e->dec_first);
    add_lstm(e->dec_lstm);
    for (auto& S : e->dec_stages) {
        add_conv_expect(e, S.resample);
        for (auto& R : S.res) { add_conv_expect(e, R.shortcut); add_conv_expect(e, R.block1); add_conv_expect(e, R.block3); }
    }
    add_conv_expect(e, e->dec_last);
    add_quantizer_expect(e);
}

// ---- weight packing -------------------------------------------------------------------------------
void choose_tiling(ConvLayer& L) {  // NOLINT
    if (L.M > 64) {
        L.BM = 128; L.BN = 128;
        if (L.small_n && L.M <= 256) { L.BM = 32; L.BN = 128; }   // too few 128x128 tiles at the bottleneck rate
        // (256 x 128 tiles -- one 8-wave workgroup per CU, the slab staged once per 256 rows -- were measured in round 2: every layer
        // slower, 18.1 vs 16.9 ms per step.  Two workgroups per CU covering each other's barriers are worth more than the halved staging.)
    } else if (L.M > 32) { L.BM = 64; L.BN = 256; }
    else { L.BM = 32; L.BN = 256; }
    // channels per K-chunk: as many as fit (a) the register-staged slab, (b) K-chunk <= 64, (c) half of the LDS
    // (two workgroups per CU).  Layers with >= 3 M tiles get their input materialised (run_conv) and therefore
    // always run the PLAIN variant: no affine tables, and the 16-element staging variant has no register pressure.
    // large strides (> 4 on the 256-column tiles, > 8 two-source): the slab of even a 2-channel chunk exceeds the staging registers.
    // Narrower N tile first, then the materialised-input (plain, 18 slots) variant, else the layer is refused at create.
    {
        auto fits = [&](bool dual) { return fc::conv_slab_fits(L.gk, L.gstride, L.dil, 2, L.BN, L.BM, dual && ceil_div_i(L.M, L.BM) < 3); };
        if (!fits(L.dual) && L.BN == 256) {
            if (L.BM == 32) L.BN = 128;
            else { L.BM = 128; L.BN = 128; }
        }
        if (!fits(L.dual)) {
            if (fits(false)) L.force_plain = true;
            else L.unsupported = true;
        }
    }
    const bool plain = ceil_div_i(L.M, L.BM) >= 3 || L.force_plain;
    const bool dual_eff = L.dual && !plain;
    // LDS budget of one workgroup.  Layers at the bottleneck frame rate with M <= 1024 have at most one workgroup per CU at the
    // benchmark shape anyway (M/128 x 2 N tiles x 16 utterances <= 256): they take the whole CU's LDS, i.e. twice as deep K
    // chunks -> half as many per-item barriers / pipeline refills for the same MFMA work (FC_SMALLN_LDS=80 restores round 1)
    static const int smalln_lds = fc::ab_knob("FC_SMALLN_LDS", 160);
    const size_t lds_budget = (size_t)((L.small_n && L.M <= 1024) ? smalln_lds : 160 / fc::conv_wgs_per_cu(L.BM)) * 1024;
    int cin_p2 = 2;
    while (cin_p2 < L.cin) cin_p2 *= 2;
    int cc = 2;
    for (;;) {
        const int n = cc * 2;
        if (n > 32 || n > cin_p2 || n * L.gk > 64) break;
        if (!fc::conv_slab_fits(L.gk, L.gstride, L.dil, n, L.BN, L.BM, dual_eff)) break;
        if (fc::conv_lds_bytes_for(L.gk, L.gstride, L.dil, n, L.BM, L.BN, L.cin, plain ? 0 : (L.dual ? 2 : 1), 0) > lds_budget) break;
        cc = n;
    }
    // stride-1 layers: row staging (16-byte loads and LDS stores) lifts the register bound on the chunk; take it when
    // it allows at least the same chunk
    static const int row_env = fc::deploy_switch("FC_ROW", 1);
    L.row = false;
    if (row_env && L.gstride == 1) {
        int best = 0;
        for (int n = 4; n <= 64 && n <= L.cin; n *= 2) {
            if (!fc::conv_row_ok(L.gk, L.gstride, L.dil, n, L.BM, L.BN, L.cin, dual_eff)) continue;
            if (fc::conv_lds_bytes_for(L.gk, L.gstride, L.dil, n, L.BM, L.BN, L.cin, plain ? 0 : (L.dual ? 2 : 1), 1) > lds_budget) continue;
            best = n;
        }
        if (best > cc || (best == cc && !plain)) { L.row = true; cc = best; }   // equal chunk + no prologue: the general
                                                                                // path's two register sets win
    }
    L.CC = cc;
    L.nchunk = ceil_div_i(L.cin, cc);
    L.Mpad = ceil_div_i(L.M, L.BM) * L.BM;
}

template <typename T>
int upload(fc_engine* e, const std::vector<T>& h, T** out) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, h.size() * sizeof(T) + 16));
    HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    e->dev_allocs.push_back(d);
    *out = (T*)d;
    return 0;
}

// gemm weight accessor: w(m, ci, tap) and bias(m)
int pack_gemm(fc_engine* e, ConvLayer& L, const std::vector<float>& wg /*[M][cin][gk]*/, const std::vector<float>& bg) {
    const int mtiles = L.Mpad / L.BM;
    const int wbuf = fc::conv_wbuf_floats(L.gk, L.CC, L.BM);   // chunk image exactly as it sits in LDS
    std::vector<float> packed((size_t)mtiles * L.nchunk * wbuf, 0.f);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int ch = 0; ch < L.nchunk; ++ch)
            for (int kk = 0; kk < L.gk; ++kk)
                for (int cl = 0; cl < L.CC; ++cl) {
                    const int ci = ch * L.CC + cl;
                    if (ci >= L.cin) continue;
                    float* img = &packed[(size_t)(mt * L.nchunk + ch) * wbuf];
                    for (int mm = 0; mm < L.BM; ++mm) {
                        const int m = mt * L.BM + mm;
                        if (m < L.M) img[fc::conv_pack_index(L.gk, L.CC, L.BM, kk, cl, mm)] = wg[((size_t)m * L.cin + ci) * L.gk + kk];
                    }
                }
    std::vector<float> bpad(L.Mpad, 0.f);
    for (int m = 0; m < L.M; ++m) bpad[m] = bg[m];
    if (upload(e, packed, &L.wt)) return 1;
    if (upload(e, bpad, &L.bias)) return 1;
    if (upload(e, fc::conv_koff_table(L.gk, L.gstride, L.dil, L.CC, L.BN, L.row ? 1 : 0), &L.koff)) return 1;
    return 0;
}

int pack_conv(fc_engine* e, ConvLayer& L) {
    const std::string inner = L.transposed ? ".convtr" : ".conv";
    std::vector<float> folded;
    if (L.wnorm) {
        // torch.nn.utils.weight_norm (conv.py:24-25): weight = v * (g / ||v||), the 2-norm taken over every dim but 0
        // (dim 0 = out channels of Conv1d, IN channels of ConvTranspose1d)
        const auto& V = e->host[L.prefix + inner + ".weight_v"].data;
        const auto& G = e->host[L.prefix + inner + ".weight_g"].data;
        const size_t d0 = L.transposed ? L.cin : L.cout, inner_n = V.size() / d0;
        folded.resize(V.size());
        for (size_t r = 0; r < d0; ++r) {
            double ss = 0.0;
            for (size_t j = 0; j < inner_n; ++j) ss += (double)V[r * inner_n + j] * (double)V[r * inner_n + j];
            const float nrm = (float)sqrt(ss);
            const float sc = G[r] / nrm;
            for (size_t j = 0; j < inner_n; ++j) folded[r * inner_n + j] = V[r * inner_n + j] * sc;
        }
    }
    const auto& W = L.wnorm ? folded : e->host[L.prefix + inner + ".weight"].data;
    const auto& Bv = e->host[L.prefix + inner + ".bias"].data;
    if (!L.transposed) {
        if (pack_gemm(e, L, W, Bv)) return 1;
        if (L.cout == 1 && L.stride == 1) {
            if (upload(e, W, &L.w_plain)) return 1;       // [1][cin][k]
            L.bias0 = Bv[0];
        }
    } else {
        // ConvTranspose1d(k = 2r, stride = r) as a 2-tap GEMM over phases: row m = co*r + p,
        // tap 0 multiplies x[i-1] with w[ci][co][p + r], tap 1 multiplies x[i] with w[ci][co][p].
        const int r = L.stride;
        if (L.k != 2 * r) return fail("ConvTranspose1d with kernel != 2*stride is not supported: " + L.prefix);
        std::vector<float> wg((size_t)L.M * L.cin * 2), bg(L.M);
        for (int co = 0; co < L.cout; ++co)
            for (int p = 0; p < r; ++p) {
                const int m = co * r + p;
                bg[m] = Bv[co];
                for (int ci = 0; ci < L.cin; ++ci) {
                    wg[((size_t)m * L.cin + ci) * 2 + 0] = W[((size_t)ci * L.cout + co) * L.k + p + r];
                    wg[((size_t)m * L.cin + ci) * 2 + 1] = W[((size_t)ci * L.cout + co) * L.k + p];
                }
            }
        if (pack_gemm(e, L, wg, bg)) return 1;
    }
    if (L.has_norm) {
        if (upload(e, e->host[L.prefix + ".norm.weight"].data, &L.gamma)) return 1;
        if (upload(e, e->host[L.prefix + ".norm.bias"].data, &L.beta)) return 1;
    }
    return 0;
}

// folded (weight-norm applied) Conv1d weight [cout][cin][k] of a layer, as pack_conv computes it
std::vector<float> folded_conv_weight(fc_engine* e, const ConvLayer& L) {
    const std::string inner = L.transposed ? ".convtr" : ".conv";
    if (!L.wnorm) return e->host[L.prefix + inner + ".weight"].data;
    const auto& V = e->host[L.prefix + inner + ".weight_v"].data;
    const auto& G = e->host[L.prefix + inner + ".weight_g"].data;
    const size_t d0 = L.transposed ? L.cin : L.cout, inner_n = V.size() / d0;
    std::vector<float> folded(V.size());
    for (size_t r = 0; r < d0; ++r) {
        double ss = 0.0;
        for (size_t j = 0; j < inner_n; ++j) ss += (double)V[r * inner_n + j] * (double)V[r * inner_n + j];
        const float sc = G[r] / (float)sqrt(ss);
        for (size_t j = 0; j < inner_n; ++j) folded[r * inner_n + j] = V[r * inner_n + j] * sc;
    }
    return folded;
}

// Thin residual blocks (C = 32 / 64): shortcut + block.1 run as ONE launch (kernels.hip 1c).  Weight images exactly as they sit
// in LDS: wsc[c][m] = W_sc[m][c][0];  wb1[kk*C + c][h] = W_b1[h][c][kk].
int pack_reshead(fc_engine* e, fc_engine::ResBlock& R) {
    static const int enable = fc::ab_knob("FC_RESHEAD", 1);
    const int C = R.shortcut.cin, hid = R.block1.cout, K = R.block1.k;
    R.fused_head = enable && R.shortcut.cout == C && R.block1.cin == C && fc::reshead_ok(C, hid, R.shortcut.k, K, R.block1.dil, R.block1.stride) &&
                   R.shortcut.has_norm == R.block1.has_norm;
    if (!R.fused_head) return 0;
    const std::vector<float> Wsc = folded_conv_weight(e, R.shortcut), Wb1 = folded_conv_weight(e, R.block1);
    std::vector<float> a((size_t)C * C), b((size_t)K * C * hid);
    for (int m = 0; m < C; ++m)
        for (int c = 0; c < C; ++c) a[(size_t)c * C + m] = Wsc[(size_t)m * C + c];
    for (int h = 0; h < hid; ++h)
        for (int c = 0; c < C; ++c)
            for (int kk = 0; kk < K; ++kk) b[((size_t)kk * C + c) * hid + h] = Wb1[((size_t)h * C + c) * K + kk];
    if (upload(e, a, &R.wsc) || upload(e, b, &R.wb1)) return 1;
    if (upload(e, e->host[R.shortcut.prefix + ".conv.bias"].data, &R.bsc)) return 1;
    if (upload(e, e->host[R.block1.prefix + ".conv.bias"].data, &R.bb1)) return 1;
    return 0;
}


// ---- STFT-domain codec: weight re-layout ---------------------------------------------------------------------------------------
// Conv2d weight [cout][C][kf][kt] -> GEMM weight [cout][a * C + ci][kt] (frequency-major layout: rows fo*sf + a are consecutive
// [C][T] blocks, so tap a of channel ci is GEMM channel a * C + ci)
int pack_conv2d(fc_engine* e, ConvLayer& L) {
    const std::vector<float> W = folded_weight_nd(e, L.prefix + ".conv", L.wnorm, (size_t)L.cout);
    const auto& Bv = e->host[L.prefix + ".conv.bias"].data;
    const int C = L.c2d, kf = L.kf, kt = L.k;
    const int cpg = C / L.groups, opg = L.cout / L.groups;        // grouped: weight [cout][C / groups][kf][kt], block-diagonal when dense
    std::vector<float> wg((size_t)L.cout * kf * C * kt, 0.f);
    for (int m = 0; m < L.cout; ++m)
        for (int cl = 0; cl < cpg; ++cl) {
            const int ci = (m / opg) * cpg + cl;
            for (int a = 0; a < kf; ++a)
                for (int b = 0; b < kt; ++b)
                    wg[((size_t)m * (kf * C) + (size_t)a * C + ci) * kt + b] = W[(((size_t)m * cpg + cl) * kf + a) * kt + b];
        }
    if (pack_gemm(e, L, wg, Bv)) return 1;
    // FC_GCONV=0: grouped layers stay on the block-diagonal dense GEMM (A/B aid)
    static const int gconv_env = fc::ab_knob("FC_GCONV", 1);
    if (gconv_env && L.groups > 1 && L.dil == 1 && fc::gconv2d_ok(cpg, opg, kf, kt, L.stride) && upload(e, W, &L.w_group)) return 1;
    if (L.cout <= 4 && L.stride == 1 && L.sf == 1 && (L.k == 3 || L.k == 5 || L.k == 7) && upload(e, wg, &L.w_plain)) return 1;      // [cout][kf * C][kt]: FMA kernel, no MFMA tile
    if (L.has_norm) {
        if (upload(e, e->host[L.prefix + ".norm.weight"].data, &L.gamma)) return 1;
        if (upload(e, e->host[L.prefix + ".norm.bias"].data, &L.beta)) return 1;
    }
    return 0;
}

// ConvTranspose2d weight [C][cout][2 sf][2 st].  Output frequency row fo = q * sf + p receives input rows fi = q (tap a = p) and
// fi = q - 1 (tap a = p + sf).  Per phase p: a ConvTranspose1d over 2 C channels (first C: row q - 1, then row q -- their order
// in the zero-haloed input buffer) with weight W1d[r * C + ci][co][bt] = W[ci][co][p + (1 - r) * sf][bt], packed like every other
// transposed layer (2-tap GEMM over the time phases).
int pack_convtr2d(fc_engine* e, const ConvLayer& S, std::vector<ConvLayer>& phases) {
    const std::vector<float> W = folded_weight_nd(e, S.prefix + ".convtr", S.wnorm, (size_t)S.c2d);
    const auto& Bv = e->host[S.prefix + ".convtr.bias"].data;
    const int C = S.c2d, cout = S.cout, kf = S.kf, kt = S.k, sf = S.sf, r = S.stride;
    const int cpg = C / S.groups, opg = cout / S.groups;          // grouped: weight [C][cout / groups][kf][kt]
    for (int p = 0; p < sf; ++p) {
        ConvLayer& L = phases[p];
        std::vector<float> wg((size_t)L.M * L.cin * 2, 0.f), bg(L.M);
        for (int co = 0; co < cout; ++co)
            for (int ph = 0; ph < r; ++ph) {
                const int m = co * r + ph;
                bg[m] = Bv[co];
                for (int rr = 0; rr < 2; ++rr)
                    for (int ci = (co / opg) * cpg; ci < (co / opg + 1) * cpg; ++ci) {
                        const int a = p + (1 - rr) * sf;
                        const float* w = &W[(((size_t)ci * opg + co % opg) * kf + a) * kt];
                        const size_t c1 = (size_t)rr * C + ci;
                        wg[((size_t)m * L.cin + c1) * 2 + 0] = w[ph + r];      // tap 0 multiplies x[i-1]
                        wg[((size_t)m * L.cin + c1) * 2 + 1] = w[ph];          // tap 1 multiplies x[i]
                    }
            }
        if (pack_gemm(e, L, wg, bg)) return 1;
    }
    ConvLayer& L0 = phases[0];
    static const int gconv_env = fc::ab_knob("FC_GCONV", 1);
    if (gconv_env && S.groups > 1 && fc::gconvtr2d_ok(cpg, opg, r)) {          // direct kernel: torch-layout weights + plain bias
        if (upload(e, W, &L0.w_group) || upload(e, Bv, &L0.w_plain)) return 1;
    }
    if (S.has_norm) {
        if (upload(e, e->host[S.prefix + ".norm.weight"].data, &L0.gamma)) return 1;
        if (upload(e, e->host[S.prefix + ".norm.bias"].data, &L0.beta)) return 1;
    }
    return 0;
}

// torch.stft / torch.istft (center, periodic Hann window of n_fft, onesided, not normalised) as conv weights over the hop-phase view
// xp[j][m] = xpad[m * hop + j]:  X[f][t] = sum_{a, j} w[n] e^{-2 pi i f n / N} xp[j][t + a],  n = a * hop + j  (zero weight for n >= N);
// ypoly[j][m] = sum_{a, c} Wi[j][c][a] S[c][m - a]  (overlap-add of the windowed inverse DFTs), as a correlation over S zero-padded by
// taps - 1 frames on both sides: tap a' = taps - 1 - a.
int pack_dft(fc_engine* e) {
    const int N = e->arch.n_fft, hop = e->arch.stft_hop, F = N / 2 + 1, taps = ceil_div_i(N, hop);
    std::vector<double> win(N);
    for (int n = 0; n < N; ++n) win[n] = 0.5 - 0.5 * cos(2.0 * M_PI * n / N);          // torch.hann_window(N) (periodic)
    // twiddles with EXACT zeros at the quarter periods: the DC and Nyquist rows then have an exactly zero imaginary part, like an FFT's
    auto tw_cos = [&](long long k) { k %= N; return (k % (N / 2)) == N / 4 ? 0.0 : cos(2.0 * M_PI * (double)k / N); };
    auto tw_sin = [&](long long k) { k %= N; return (k % (N / 2)) == 0 ? 0.0 : sin(2.0 * M_PI * (double)k / N); };
    {
        ConvLayer& L = e->stft;
        std::vector<float> wg((size_t)2 * F * hop * taps, 0.f), bg(2 * F, 0.f);
        for (int f = 0; f < F; ++f)
            for (int j = 0; j < hop; ++j)
                for (int a = 0; a < taps; ++a) {
                    const int n = a * hop + j;
                    if (n >= N) continue;
                    const long long k = (long long)f * n;
                    wg[((size_t)f * hop + j) * taps + a] = (float)(win[n] * tw_cos(k));
                    wg[((size_t)(F + f) * hop + j) * taps + a] = tw_sin(k) == 0.0 ? 0.f : (float)(-win[n] * tw_sin(k));
                }
        if (pack_gemm(e, L, wg, bg)) return 1;
    }
    {
        ConvLayer& L = e->istft;
        std::vector<float> wg((size_t)hop * 2 * F * taps, 0.f), bg(hop, 0.f);
        for (int j = 0; j < hop; ++j)
            for (int f = 0; f < F; ++f)
                for (int a = 0; a < taps; ++a) {
                    const int n = a * hop + j;
                    if (n >= N) continue;
                    const double cf = (f == 0 || f == N / 2) ? 1.0 : 2.0;
                    const long long k = (long long)f * n;
                    const int ap = taps - 1 - a;
                    wg[((size_t)j * 2 * F + f) * taps + ap] = (float)(win[n] * cf * tw_cos(k) / N);
                    wg[((size_t)j * 2 * F + F + f) * taps + ap] = tw_sin(k) == 0.0 ? 0.f : (float)(-win[n] * cf * tw_sin(k) / N);
                }
        if (pack_gemm(e, L, wg, bg)) return 1;
    }
    std::vector<float> w2(N);
    for (int n = 0; n < N; ++n) w2[n] = (float)((float)win[n] * (float)win[n]);
    if (upload(e, w2, &e->win2)) return 1;
    return 0;
}

int pack_lstm(fc_engine* e, LstmBlock& lb) {
    const int H = lb.H;
    if (H == 0) return 0;
    if (H % 16 != 0) return fail("LSTM width must be a multiple of 16");
    for (int l = 0; l < (int)lb.layers.size(); ++l) {
        const std::string sfx = "_l" + std::to_string(l);
        const auto& Wih = e->host[lb.prefix + ".weight_ih" + sfx].data;
        const auto& Whh = e->host[lb.prefix + ".weight_hh" + sfx].data;
        const auto& bih = e->host[lb.prefix + ".bias_ih" + sfx].data;
        const auto& bhh = e->host[lb.prefix + ".bias_hh" + sfx].data;
        // permuted row m' = blk*16 + u*4 + gate  <->  original row gate*H + blk*4 + u   (gate order i,f,g,o)
        std::vector<float> wih_p((size_t)4 * H * H), whh_p((size_t)4 * H * H), b_p(4 * H);
        for (int mp = 0; mp < 4 * H; ++mp) {
            const int blk = mp / 16, u = (mp % 16) / 4, gate = mp % 4;
            const int orig = gate * H + blk * 4 + u;
            memcpy(&wih_p[(size_t)mp * H], &Wih[(size_t)orig * H], H * sizeof(float));
            memcpy(&whh_p[(size_t)mp * H], &Whh[(size_t)orig * H], H * sizeof(float));
            b_p[mp] = bih[orig] + bhh[orig];
        }
        LstmLayer& L = lb.layers[l];
        if (pack_gemm(e, L.inproj, wih_p, b_p)) return 1;
        if (upload(e, whh_p, &L.whh)) return 1;
        if (l >= 1) {
            std::vector<float> cat((size_t)4 * H * 2 * H);
            for (int mp = 0; mp < 4 * H; ++mp) {
                memcpy(&cat[(size_t)mp * 2 * H], &wih_p[(size_t)mp * H], H * sizeof(float));
                memcpy(&cat[(size_t)mp * 2 * H + H], &whh_p[(size_t)mp * H], H * sizeof(float));
            }
            if (upload(e, cat, &L.wcat)) return 1;
            if (upload(e, b_p, &L.bperm)) return 1;
        }
    }
    return 0;
}

// ---- execution ------------------------------------------------------------------------------------
struct ConvGeom { int Tout, padL, padR, count_T; };

// SConv1d.forward padding arithmetic (conv.py:243-258, get_extra_padding_for_conv1d :57-64)
ConvGeom conv_geom(const ConvLayer& L, int T) {
    ConvGeom g;
    if (L.valid) { g.padL = 0; g.padR = 0; g.Tout = T - L.k + 1; g.count_T = g.Tout; return g; }
    if (L.zpadL >= 0) { g.padL = L.zpadL; g.padR = L.zpadR; g.Tout = T + g.padL + g.padR - L.k + 1; g.count_T = g.Tout; return g; }
    if (!L.transposed) {
        const int pt = (L.k - 1) * L.dil - (L.stride - 1);      // padding_total (conv.py:247)
        const int num = T - L.k + pt;
        const int nfr = num >= 0 ? ceil_div_i(num, L.stride) : -((-num) / L.stride);   // ceil(n_frames) - 1
        const int ideal = nfr * L.stride + (L.k - pt);
        const int extra = ideal - T;
        if (L.causal) { g.padL = pt; g.padR = extra; }          // all fixed padding on the left (conv.py:249-251)
        else if (L.extra_left) { g.padR = pt / 2; g.padL = pt - pt / 2 + extra; }   // SConv2d, time axis (conv.py:376-377)
        else { g.padR = pt / 2 + extra; g.padL = pt - pt / 2; }
        // what nn.Conv1d then produces on the padded input (the reference's frame formula above ignores dilation; with
        // dilation 1 this equals nfr + 1)
        g.Tout = (T + g.padL + g.padR - ((L.k - 1) * L.dil + 1)) / L.stride + 1;
        g.count_T = g.Tout;
    } else {
        g.padL = 1; g.padR = 1;
        g.Tout = T * L.stride;           // trimmed length
        g.count_T = (T + 1) * L.stride;  // GroupNorm sees the untrimmed output (conv.py:287-303)
    }
    return g;
}

static bool out_override_blocks_xq(const ConvLayer& L) { return L.kf != 1 || L.valid || L.cout <= 4; }   // 2-D / STFT GEMMs / few-output FMA layers
Act run_conv(fc_engine* e, Ctx& cx, const ConvLayer& L, fc::Src s0, fc::Src s1, int elu, int Tin,
             float* out_override = nullptr, long long sB = 0, long long sM = 0, long long sT = 0) {
    const ConvGeom g = conv_geom(L, Tin);
    // Layers with several M tiles would re-apply the fused prologue (GroupNorm affine, residual add, ELU) once per
    // M tile; for those (deep, short tensors) it is cheaper to materialise the activated input once and stream it.
    const bool has_prologue = (s0.used & 2) || s1.used || elu;
    int xq_Tp = 0;
    if ((L.Mpad / L.BM >= 3 || L.force_plain) && has_prologue) {
        // the conv kernel's quad layout stages such an input by DMA when it is materialised with 4 channels interleaved and its padding in
        // place (kernels.hip combine_xq_kernel / conv_kernel.h MODE 5); else the plain [B][C][T] tensor and the register-staged PLAIN form
        const bool xq = !s0.div && !out_override_blocks_xq(L) && fc::conv_xq_ok(L.cin, L.CC, L.gk, L.gstride, L.dil, L.BM, L.BN, L.row ? 1 : 0);
        const int padL_x = L.transposed ? 1 : g.padL, padR_x = L.transposed ? 1 : g.padR;
        const int Tp = padL_x + Tin + padR_x;
        float* tmp = xq ? cx.alloc<float>(fc::conv_xq_floats(cx.B, L.cin, Tp, L.BN, L.gstride, L.gk, L.dil))
                        : cx.alloc<float>((size_t)cx.B * L.cin * Tin);
        cx.launches++;
        if (!cx.dry && !cx.err) {
            hipError_t er0 = xq ? fc::launch_combine_xq(s0, s1, elu, e->arch.elu_alpha, cx.B, L.cin, Tin, padL_x, padR_x,
                                                        (L.transposed || L.zpadL >= 0) ? 1 : 0, tmp, cx.st)
                                : fc::launch_combine(s0, s1, elu, e->arch.elu_alpha, nullptr, cx.B, L.cin, Tin, Tin, tmp,
                                                     (long long)L.cin * Tin, Tin, 1, cx.st);
            if (er0 != hipSuccess) { cx.err = 1; g_err = std::string("combine launch failed: ") + hipGetErrorString(er0); }
        }
        s0 = fc::Src(); s0.ptr = tmp; s0.used = 1;
        s1 = fc::Src();
        elu = 0;
        if (xq) xq_Tp = Tp;
    }
    fc::ConvLaunch c;
    c.s0 = s0; c.s1 = s1; c.elu = elu; c.alpha = e->arch.elu_alpha;
    c.wt = L.wt; c.bias = L.bias; c.koff = L.koff;
    c.B = cx.B; c.Cin = L.cin; c.Tin = Tin; c.M = L.M;
    c.k = L.gk; c.stride = L.gstride; c.dil = L.dil; c.padL = g.padL; c.padR = g.padR;
    c.BM = L.BM; c.BN = L.BN; c.CC = L.CC; c.nchunk = L.nchunk; c.row = L.row ? 1 : 0;
    c.w_plain = L.w_plain; c.bias_host0 = L.bias0;
    c.xq_Tp = xq_Tp;
    Act out;
    out.C = L.cout; out.T = g.Tout;
    if (L.transposed) {
        c.Tout = Tin + 1; c.pad_zero = 1;
        // unpad1d: non-causal trims left = r - r/2, right = r/2; causal (trim_right_ratio 1) trims everything on the right
        c.up_r = L.stride; c.trimL = L.causal ? 0 : L.stride - L.stride / 2; c.Tfinal = g.Tout;
    } else {
        c.Tout = g.Tout;
        if (L.zpadL >= 0) c.pad_zero = 1;
    }
    if (out_override) {
        out.raw = out_override;
        c.out_sB = sB; c.out_sM = sM; c.out_sT = sT;
    } else {
        out.raw = cx.alloc<float>((size_t)cx.B * L.cout * g.Tout);
        c.out_sB = (long long)L.cout * g.Tout; c.out_sM = g.Tout; c.out_sT = 1;
    }
    c.out = out.raw;
    const int nblk = fc::conv_nblk(c);
    if (L.has_norm) {
        c.partials = cx.alloc<double>((size_t)cx.B * nblk * 2);
        out.aff = cx.alloc<float>((size_t)cx.B * L.cout * 2);
        out.normed = true;
    }
    // accounting (algorithmic: real channel counts, every operand touched once)
    cx.conv_flops += 2.0 * cx.B * (double)L.M * L.cin * L.gk * c.Tout;
    cx.conv_bytes += 4.0 * cx.B * ((double)L.cin * Tin * (s1.used ? 2 : 1) + (double)L.cout * g.Tout);
    cx.launches += L.has_norm ? 2 : 1;
    cx.conv_launches += 1;
    if (cx.dry || cx.err) return out;
    const double fl = 2.0 * cx.B * (double)L.M * L.cin * L.gk * c.Tout;
    const double by = 4.0 * cx.B * ((double)L.cin * Tin * (s1.used ? 2 : 1) + (double)L.cout * g.Tout);
    hipError_t er;
    {
        int cls = 0;
        if (e->profiling) {
            int mode = 0, nu = 0;
            int row = 0;
            fc::conv_variant(c, &mode, &nu, &row);
            char nm[80];
            snprintf(nm, sizeof(nm), "conv_mfma_kernel<%d, %d, %d, %d, %d, %d, %s, %s>", L.BM, L.BN, L.BM >= 128 ? 2 : 1,
                     L.BM >= 128 ? 2 : 4, mode, nu, (row & 1) ? "true" : "false", (row & 2) ? "true" : "false");
            if (fc::conv_cout1_ok(c))
                fc::conv_fewout_name(c, nm, sizeof(nm));
            cls = e->prof_class(nm);
        }
        ProfSpan sp(e, cx, cls, fl, by);
        er = fc::launch_conv(c, cx.st);
    }
    if (er != hipSuccess) { cx.err = 1; g_err = "conv launch failed (" + L.prefix + "): " + hipGetErrorString(er) +
                " (invalid value = tiling / LDS limits, or an utterance longer than 2^32 / (4 * channels per chunk) samples)"; return out; }
    if (L.has_norm) {
        er = fc::launch_gn_finalize(c.partials, nblk, (double)L.cout * g.count_T, L.gamma, L.beta, L.cout, e->arch.gn_eps,
                                    cx.B, out.aff, cx.st);
        if (er != hipSuccess) { cx.err = 1; g_err = std::string("gn_finalize launch failed: ") + hipGetErrorString(er); }
    }
    return out;
}

// Src.used: bit 0 = present, bit 1 = carries an affine / divisor (planning flags, valid when pointers are null)
inline fc::Src src_of(const Act& a) { fc::Src s; s.ptr = a.raw; s.aff = a.aff; s.used = a.normed ? 3 : 1; return s; }

// SLSTM.forward (lstm.py:22-28) without the skip; returns plain y [B][H][T].
// Layer wavefront: one launch per "diagonal" s advances every layer l by its timestep s - l.
Act run_lstm(fc_engine* e, Ctx& cx, const LstmBlock& lb, const Act& in, int T) {
    const int H = lb.H, B = cx.B, L = (int)lb.layers.size();
    float* xproj = cx.alloc<float>((size_t)T * B * 4 * H);
    run_conv(e, cx, lb.layers[0].inproj, src_of(in), fc::Src(), 0, T, xproj, (long long)4 * H, 1, (long long)B * 4 * H);
    static const int persist_env = fc::deploy_switch("FC_LSTM_PERSIST", 1);
    bool persist = false;
    if (persist_env && e->lstm_persist_ok) {
        const int key = B * 4096 + H;            // the occupancy query is cached per (B, H)
        if (e->persist_checked_B != key) { e->persist_checked_val = fc::lstm_persist_supported(B, H, L, e->device); e->persist_checked_B = key; }
        persist = e->persist_checked_val;
    }
    // persistent kernel: barrier words + hidden-state history; per-step launches: h [L][2][B][H], c [L][B][H]
    float* state = cx.alloc<float>(persist ? fc::lstm_persist_state_floats(B, H, T) : (size_t)3 * L * B * H);
    Act y;
    y.C = H; y.T = T;
    y.raw = cx.alloc<float>((size_t)B * H * T);
    cx.lstm_flops += 2.0 * B * (double)T * 4 * H * H * (2 * L - 1);
    cx.launches += persist ? 2 : T + L;
    if (!cx.dry && !cx.err) {
        // algorithmic bytes per SURVEY.md 8d: the recurrent weights read ONCE (they stay resident in registers) + the x-projection in + y out;
        // the per-step launches re-stream the weights, which is traffic, not algorithm
        const double lstm_bytes = 4.0 * (4.0 * H * H * (2 * L - 1) + (double)T * B * 4.0 * H + (double)B * H * T);
        ProfSpan sp(e, cx, e->profiling ? e->prof_class(persist ? kLstmPersistClass : kLstmWaveClass) : 0, 2.0 * B * (double)T * 4 * H * H * (2 * L - 1), lstm_bytes);
        hipError_t er = fc::launch_zero_fill(state, persist ? fc::lstm_persist_clear_floats(B, H) : (size_t)3 * L * B * H, cx.st);
        const float* w[FC_LSTM_MAX_LAYERS] = {nullptr};
        const float* bias[FC_LSTM_MAX_LAYERS] = {nullptr};
        for (int l = 0; l < L; ++l) { w[l] = l == 0 ? lb.layers[0].whh : lb.layers[l].wcat; bias[l] = lb.layers[l].bperm; }
        if (persist) {
            if (er == hipSuccess) er = fc::launch_lstm_persist(w[0], w[1], bias[1], xproj, state, y.raw, B, H, T, e->status_dev, cx.st);
        } else {
            float *h = state, *c = state + (size_t)2 * L * B * H;
            for (int s = 0; s < T + L - 1 && er == hipSuccess; ++s)
                er = fc::launch_lstm_wave(w, bias, xproj, h, c, y.raw, B, H, T, L, s, cx.st);
        }
        if (er != hipSuccess) { cx.err = 1; g_err = std::string("lstm step failed: ") + hipGetErrorString(er); }
    }
    return y;
}

// SEANetResnetBlock (seanet_encoder.py:16-61): returns the two raw branches whose GroupNorm'd sum is the output
// The block's input is one tensor (first block of a stage) or the pending sum of the previous block's two branches.
// shortcut(x) and block.1(ELU(x)) of a thin residual block from ONE staging of x (kernels.hip 1c); same outputs / statistics
// contract as two run_conv() calls
void run_reshead(fc_engine* e, Ctx& cx, const fc_engine::ResBlock& R, fc::Src a0, fc::Src a1, int T, Act* sc, Act* b1) {
    const int C = R.shortcut.cin, hid = R.block1.cout;
    const ConvGeom g = conv_geom(R.block1, T);              // stride 1: Tout == T
    sc->C = C; sc->T = T; b1->C = hid; b1->T = T;
    sc->raw = cx.alloc<float>((size_t)cx.B * C * T);
    b1->raw = cx.alloc<float>((size_t)cx.B * hid * T);
    fc::ResHeadLaunch c;
    c.s0 = a0; c.s1 = a1; c.wsc = R.wsc; c.wb1 = R.wb1; c.bsc = R.bsc; c.bb1 = R.bb1;
    c.out_sc = sc->raw; c.out_b1 = b1->raw;
    c.B = cx.B; c.C = C; c.T = T; c.k = R.block1.k; c.dil = R.block1.dil; c.padL = g.padL; c.padR = g.padR;
    c.alpha = e->arch.elu_alpha;
    const int nblk = fc::reshead_ntiles(T);
    if (R.shortcut.has_norm) {
        c.part_sc = cx.alloc<double>((size_t)cx.B * nblk * 2);
        c.part_b1 = cx.alloc<double>((size_t)cx.B * nblk * 2);
        sc->aff = cx.alloc<float>((size_t)cx.B * C * 2); sc->normed = true;
        b1->aff = cx.alloc<float>((size_t)cx.B * hid * 2); b1->normed = true;
    }
    const double fl = 2.0 * cx.B * (double)T * ((double)C * C + (double)hid * C * R.block1.k);
    const double by = 4.0 * cx.B * (double)T * ((double)C * (a1.used ? 2 : 1) + C + hid);      // x read ONCE, both outputs written
    cx.conv_flops += fl; cx.conv_bytes += by;
    cx.launches += R.shortcut.has_norm ? 3 : 1;
    cx.conv_launches += 1;
    if (cx.dry || cx.err) return;
    hipError_t er;
    {
        int cls = 0;
        if (e->profiling) {
            char nm[64];
            snprintf(nm, sizeof(nm), "reshead_kernel<%d, %d, %s>", C, R.block1.k, a1.ptr ? "true" : "false");
            cls = e->prof_class(nm);
        }
        ProfSpan sp(e, cx, cls, fl, by);
        er = fc::launch_reshead(c, cx.st);
    }
    if (er != hipSuccess) { cx.err = 1; g_err = "fused res-block head launch failed (" + R.shortcut.prefix + "): " + hipGetErrorString(er); return; }
    if (R.shortcut.has_norm) {
        er = fc::launch_gn_finalize(c.part_sc, nblk, (double)C * T, R.shortcut.gamma, R.shortcut.beta, C, e->arch.gn_eps, cx.B, sc->aff, cx.st);
        if (er == hipSuccess)
            er = fc::launch_gn_finalize(c.part_b1, nblk, (double)hid * T, R.block1.gamma, R.block1.beta, hid, e->arch.gn_eps, cx.B, b1->aff, cx.st);
        if (er != hipSuccess) { cx.err = 1; g_err = std::string("gn_finalize launch failed: ") + hipGetErrorString(er); }
    }
}

// SEANetResnetBlock (seanet_encoder.py:16-61): returns the two raw branches whose GroupNorm'd sum is the output
// The block's input is one tensor (first block of a stage) or the pending sum of the previous block's two branches.
void run_resblocks(fc_engine* e, Ctx& cx, const fc_engine::Stage& S, fc::Src a0, fc::Src a1, int T, Act* sc, Act* b3) {
    for (const auto& R : S.res) {
        Act b1;
        if (R.fused_head && !a0.div) {
            run_reshead(e, cx, R, a0, a1, T, sc, &b1);
        } else {
            *sc = run_conv(e, cx, R.shortcut, a0, a1, 0, T);
            b1 = run_conv(e, cx, R.block1, a0, a1, 1, T);
        }
        *b3 = run_conv(e, cx, R.block3, src_of(b1), fc::Src(), 1, b1.T);
        a0 = src_of(*sc); a1 = src_of(*b3);
    }
}

// SEANetEncoder.forward: wav [B][T] (optionally divided by scale[b]) -> last conv (raw + affine), T -> Tf
Act run_encoder(fc_engine* e, Ctx& cx, const float* wav, int T, const float* scale) {
    fc::Src s; s.ptr = wav; s.div = scale; s.used = e->arch.audio_normalize ? 3 : 1;
    Act x = run_conv(e, cx, e->enc_first, s, fc::Src(), 0, T);
    for (auto& S : e->enc_stages) {
        Act sc, b3;
        run_resblocks(e, cx, S, src_of(x), fc::Src(), x.T, &sc, &b3);
        x = run_conv(e, cx, S.resample, src_of(sc), src_of(b3), 1, sc.T);
    }
    if (e->enc_lstm.H) {
        Act y = run_lstm(e, cx, e->enc_lstm, x, x.T);
        if (e->arch.lstm_skip) return run_conv(e, cx, e->enc_last, src_of(y), src_of(x), 1, x.T);
        return run_conv(e, cx, e->enc_last, src_of(y), fc::Src(), 1, x.T);
    }
    return run_conv(e, cx, e->enc_last, src_of(x), fc::Src(), 1, x.T);
}

// SEANetDecoder.forward: z [B][D][Tf] plain -> last conv (raw [B][1][Tf*hop] + affine)
Act run_decoder(fc_engine* e, Ctx& cx, const float* z_bdt, int Tf) {
    fc::Src s; s.ptr = z_bdt; s.used = 1;
    Act x = run_conv(e, cx, e->dec_first, s, fc::Src(), 0, Tf);
    fc::Src a0 = src_of(x), a1;
    if (e->dec_lstm.H) {
        Act y = run_lstm(e, cx, e->dec_lstm, x, x.T);
        a0 = src_of(y);
        if (e->arch.lstm_skip) a1 = src_of(x);
    }
    int T = Tf;
    for (auto& S : e->dec_stages) {
        Act up = run_conv(e, cx, S.resample, a0, a1, 1, T);
        Act sc, b3;
        run_resblocks(e, cx, S, src_of(up), fc::Src(), up.T, &sc, &b3);
        a0 = src_of(sc); a1 = src_of(b3);
        T = up.T;
    }
    return run_conv(e, cx, e->dec_last, a0, a1, 1, T);
}


// FreqCodec: the direct kernels and the materialisation passes write the reflected / zero halo rows of their outputs themselves (round 4;
// FC_HALO_FUSE=0: a halo_rows launch behind every producer, as before -- A / B aid)
static bool halo_fuse_on() {
    static const int v = fc::ab_knob("FC_HALO_FUSE", 1);
    return v != 0;
}

// ---- STFT-domain codec: execution over frequency-major activations --------------------------------------------------------------
struct Act2 {              // raw [B][F + 2*halo][C][T] + pending GroupNorm affine [B][C] (null = final)
    float* buf = nullptr;
    float* aff = nullptr;
    int C = 0, F = 0, T = 0, halo = 0;
    bool normed = false;
};
// fc_engine::halo2 (set by build_plan_2d): frequency halo rows of every 2-D activation that a kf > 1 conv may read = the largest one-sided
// frequency padding of the net's convs (7x7: 3, 3x3: 1, strided 2 fr rows with stride fr: ceil(fr / 2))

// SConv2d.forward (conv.py:342-381) as ONE launch of the 1-D implicit-GEMM kernel over B * Fo virtual utterances
Act2 run_conv2d(fc_engine* e, Ctx& cx, const ConvLayer& L, Act2 x0, const Act2* x1p, int elu, int out_halo) {
    const int B = cx.B, C = L.c2d, kf = L.kf, sf = L.sf;
    Act2 x1 = x1p ? *x1p : Act2();
    bool dual = x1p != nullptr;
    const int tot_f = (kf - 1) - (sf - 1), f_after = tot_f / 2, f_before = tot_f - f_after;     // no extra padding on the frequency axis
    const int Fo = (x0.F + tot_f - kf) / sf + 1;
    const ConvGeom g = conv_geom(L, x0.T);
    // layers with several M tiles: materialise the activated input once (as run_conv does), frequency-major with the same halo
    if (L.w_group) {     // grouped conv with 2 / 4 channels per group: direct FMA kernel, prologue fused, no GEMM
        // the strided 8-row layers read every input sample from 2 output rows x ~1.25 lanes with a two-source prologue (11 VALU per read):
        // activate once instead (FC_GCONV_MAT=0: in-kernel prologue)
        static const int gmat = fc::ab_knob("FC_GCONV_MAT", 1);
        if (gmat && kf >= 4 && dual) {
            Act2 m;
            m.C = C; m.F = x0.F; m.T = x0.T; m.halo = x0.halo;
            m.buf = cx.alloc<float>((size_t)B * (m.F + 2 * m.halo) * C * m.T);
            const bool fuse_halo = halo_fuse_on() && m.F > m.halo;      // the pass writes its own reflected halo rows
            cx.launches += fuse_halo ? 1 : 2;
            if (!cx.dry && !cx.err) {
                hipError_t er = fc::launch_combine2d(x0.buf, x0.aff, x0.halo, x1.buf, x1.aff, x1.halo, elu, e->arch.elu_alpha, B, m.F, C, m.T, m.buf,
                                                     m.halo, cx.st, fuse_halo ? 1 : 0);
                if (er == hipSuccess && !fuse_halo) er = fc::launch_halo_rows(m.buf, B, m.F, m.halo, C, m.T, 0, cx.st);
                if (er != hipSuccess) { cx.err = 1; g_err = std::string("combine2d launch failed: ") + hipGetErrorString(er); }
            }
            x0 = m; x1 = Act2(); elu = 0; dual = false;
        }
        Act2 o;
        o.C = L.cout; o.F = Fo; o.T = g.Tout; o.halo = out_halo;
        o.buf = cx.alloc<float>((size_t)B * (Fo + 2 * out_halo) * L.cout * g.Tout);
        if (!cx.dry && (x0.halo < f_before || x0.halo < f_after || (x1.buf && x1.halo != x0.halo))) {
            cx.err = 1; g_err = "internal: frequency halo too small for " + L.prefix; return o;
        }
        const long long rowsz = (long long)C * x0.T, orow = (long long)L.cout * g.Tout;
        fc::GConvLaunch c;
        c.src0 = cx.dry ? nullptr : x0.buf + (long long)(x0.halo - f_before) * rowsz; c.aff0 = x0.aff;
        if (dual) { c.src1 = cx.dry ? nullptr : x1.buf + (long long)(x1.halo - f_before) * rowsz; c.aff1 = x1.aff; }
        c.w = L.w_group; c.bias = L.bias;
        c.out = cx.dry ? nullptr : o.buf + (long long)out_halo * orow;
        c.B = B; c.C = C; c.M = L.cout; c.G = L.groups; c.Tin = x0.T; c.Tout = g.Tout; c.Fo = Fo; c.kf = kf; c.kt = L.k; c.sf = sf; c.st = L.stride;
        c.padL = g.padL; c.padR = g.padR; c.elu = elu; c.alpha = e->arch.elu_alpha;
        c.guard_lo = cx.base; c.guard_hi = cx.base ? cx.base + cx.cap : nullptr;      // both sources are buffers of THIS workspace (Ctx::alloc)
        c.in_sB = (long long)(x0.F + 2 * x0.halo) * rowsz; c.in_sF = rowsz;
        c.out_sB = (long long)(Fo + 2 * out_halo) * orow; c.out_sF = orow;
        const bool fuse_halo = halo_fuse_on() && fc::gconv2d_fuses_halo(kf, L.k, L.stride, Fo, out_halo);
        c.out_halo = fuse_halo ? out_halo : 0;
        const int nblk = fc::gconv2d_nblk(g.Tout, Fo, L.groups, kf);
        c.partials = L.has_norm ? cx.alloc<double>((size_t)B * nblk * 2) : nullptr;       // weight_norm nets: no statistics, no affine
        o.aff = L.has_norm ? cx.alloc<float>((size_t)B * L.cout * 2) : nullptr;
        o.normed = L.has_norm;
        const double fl = 2.0 * B * Fo * (double)L.cout * (C / L.groups) * kf * L.k * g.Tout;
        const double by = 4.0 * B * ((double)C * x0.F * x0.T * (dual ? 2 : 1) + (double)L.cout * Fo * g.Tout);
        cx.conv_flops += fl; cx.conv_bytes += by;
        cx.launches += (out_halo && !fuse_halo) ? 3 : 2; cx.conv_launches += 1;
        if (cx.dry || cx.err) return o;
        hipError_t er;
        {
            int cls = 0;
            if (e->profiling) {
                char nm[64];
                snprintf(nm, sizeof(nm), "gconv2d_kernel<%d, %d, %d, %d, %d, %s>", C / L.groups, L.cout / L.groups, kf, L.k, L.stride, dual ? "true" : "false");
                cls = e->prof_class(nm);
            }
            ProfSpan sp(e, cx, cls, fl, by);
            er = fc::launch_gconv2d(c, cx.st);
        }
        if (er == hipSuccess && L.has_norm)
            er = fc::launch_gn_finalize(c.partials, nblk, (double)L.cout * Fo * g.count_T, L.gamma, L.beta, L.cout, e->arch.gn_eps, B, o.aff, cx.st);
        if (er == hipSuccess && out_halo && !fuse_halo) er = fc::launch_halo_rows(o.buf, B, Fo, out_halo, L.cout, g.Tout, 0, cx.st);
        if (er != hipSuccess) { cx.err = 1; g_err = "grouped 2-D conv launch failed (" + L.prefix + "): " + hipGetErrorString(er); }
        return o;
    }
    // the few-output FMA kernel re-stages every input row for each of the kf output rows that read it: activate once instead
    static const int few_mat = fc::ab_knob("FC_FEWOUT_MAT", 1);
    const bool few_out = few_mat && L.cout <= 4 && L.stride == 1 && sf == 1 && (L.k == 3 || L.k == 5 || L.k == 7) && kf > 1;   // = pack_conv2d's w_plain rule
    if ((L.Mpad / L.BM >= 3 || few_out || L.force_plain) && (x0.normed || dual || elu)) {
        Act2 m;
        m.C = C; m.F = x0.F; m.T = x0.T; m.halo = x0.halo;
        m.buf = cx.alloc<float>((size_t)B * (m.F + 2 * m.halo) * C * m.T);
        const bool fuse_halo = halo_fuse_on() && m.F > m.halo;
        cx.launches += fuse_halo ? 1 : 2;
        if (!cx.dry && !cx.err) {
            hipError_t er = fc::launch_combine2d(x0.buf, x0.aff, x0.halo, dual ? x1.buf : nullptr, dual ? x1.aff : nullptr, x1.halo, elu,
                                                 e->arch.elu_alpha, B, m.F, C, m.T, m.buf, m.halo, cx.st, fuse_halo ? 1 : 0);
            if (er == hipSuccess && !fuse_halo) er = fc::launch_halo_rows(m.buf, B, m.F, m.halo, C, m.T, 0, cx.st);
            if (er != hipSuccess) { cx.err = 1; g_err = std::string("combine2d launch failed: ") + hipGetErrorString(er); }
        }
        x0 = m; x1 = Act2(); elu = 0; dual = false;
    }
    const bool dual_eff = dual;
    Act2 o;
    o.C = L.cout; o.F = Fo; o.T = g.Tout; o.halo = out_halo;
    o.buf = cx.alloc<float>((size_t)B * (Fo + 2 * out_halo) * L.cout * g.Tout);
    if (!cx.dry && (x0.halo < f_before || x0.halo < f_after || (x1.buf && x1.halo != x0.halo))) {
        cx.err = 1; g_err = "internal: frequency halo too small for " + L.prefix; return o;
    }
    fc::ConvLaunch c;
    const long long rowsz = (long long)C * x0.T;
    c.s0.ptr = cx.dry ? nullptr : x0.buf + (long long)(x0.halo - f_before) * rowsz;
    c.s0.aff = x0.aff; c.s0.used = x0.normed ? 3 : 1;
    if (dual_eff) { c.s1.ptr = cx.dry ? nullptr : x1.buf + (long long)(x1.halo - f_before) * rowsz; c.s1.aff = x1.aff; c.s1.used = x1.normed ? 3 : 1; }
    c.elu = elu; c.alpha = e->arch.elu_alpha;
    c.wt = L.wt; c.bias = L.bias; c.koff = L.koff;
    c.B = B * Fo; c.Cin = L.cin; c.Tin = x0.T; c.M = L.M; c.Tout = g.Tout;
    c.k = L.gk; c.stride = L.gstride; c.dil = L.dil; c.padL = g.padL; c.padR = g.padR;
    c.BM = L.BM; c.BN = L.BN; c.CC = L.CC; c.nchunk = L.nchunk; c.row = L.row ? 1 : 0;
    c.Fo = Fo; c.affC = C; c.w_plain = L.w_plain;
    c.in_sB0 = (long long)(x0.F + 2 * x0.halo) * rowsz; c.in_sB1 = (long long)sf * rowsz;
    const long long orow = (long long)L.cout * g.Tout;
    c.out = cx.dry ? nullptr : o.buf + (long long)out_halo * orow;
    c.out_sB = (long long)(Fo + 2 * out_halo) * orow; c.out_sF = orow; c.out_sM = g.Tout; c.out_sT = 1;
    const int nblk = fc::conv_nblk(c);
    c.partials = L.has_norm ? cx.alloc<double>((size_t)B * Fo * nblk * 2) : nullptr;
    o.aff = L.has_norm ? cx.alloc<float>((size_t)B * L.cout * 2) : nullptr;
    o.normed = L.has_norm;
    const double fl = 2.0 * B * Fo * (double)L.M * L.cin * L.gk * g.Tout;
    const double by = 4.0 * B * ((double)C * x0.F * x0.T * (dual_eff ? 2 : 1) + (double)L.cout * Fo * g.Tout);
    cx.conv_flops += fl; cx.conv_bytes += by;
    cx.launches += out_halo ? 3 : 2; cx.conv_launches += 1;
    if (cx.dry || cx.err) return o;
    hipError_t er;
    {
        int cls = 0;
        if (e->profiling) {
            int mode = 0, nu = 0, row = 0;
            fc::conv_variant(c, &mode, &nu, &row);
            char nm[80];
            snprintf(nm, sizeof(nm), "conv_mfma_kernel<%d, %d, %d, %d, %d, %d, %s, %s>", L.BM, L.BN, L.BM >= 128 ? 2 : 1, L.BM >= 128 ? 2 : 4, mode, nu,
                     (row & 1) ? "true" : "false", (row & 2) ? "true" : "false");
            if (fc::conv_cout1_ok(c))
                fc::conv_fewout_name(c, nm, sizeof(nm));
            cls = e->prof_class(nm);
        }
        ProfSpan sp(e, cx, cls, fl, by);
        er = fc::launch_conv(c, cx.st);
    }
    if (er == hipSuccess && L.has_norm)
        er = fc::launch_gn_finalize(c.partials, nblk * Fo, (double)L.cout * Fo * g.count_T, L.gamma, L.beta, L.cout, e->arch.gn_eps, B, o.aff, cx.st);
    if (er == hipSuccess && out_halo) er = fc::launch_halo_rows(o.buf, B, Fo, out_halo, L.cout, g.Tout, 0, cx.st);
    if (er != hipSuccess) { cx.err = 1; g_err = "2-D conv launch failed (" + L.prefix + "): " + hipGetErrorString(er); }
    return o;
}

// SEANetResnetBlock2d.forward (seanet_encoder.py:239-240) over the blocks of a stage: returns the two raw branches of the last block
void run_resblocks2d(fc_engine* e, Ctx& cx, const fc_engine::Stage& S, Act2 a0, const Act2* a1, Act2* sc, Act2* b3) {
    Act2 prev1;
    for (const auto& R : S.res) {
        *sc = run_conv2d(e, cx, R.shortcut, a0, a1, 0, e->halo2);
        Act2 b1 = run_conv2d(e, cx, R.block1, a0, a1, 1, 0);
        *b3 = run_conv2d(e, cx, R.block3, b1, nullptr, 1, e->halo2);
        a0 = *sc; prev1 = *b3; a1 = &prev1;
    }
}

// SConvTranspose2d.forward (conv.py:408-447): sf launches (one per output-frequency phase) of the transposed 1-D GEMM over the
// materialised, zero-haloed input; GroupNorm statistics over the UNTRIMMED (Fin + 1) sf x (Tin + 1) st output
Act2 run_convtr2d(fc_engine* e, Ctx& cx, const ConvLayer& S, const std::vector<ConvLayer>& phases, const Act2& x0, const Act2* x1, bool last,
                  int out_halo) {
    const int B = cx.B, C = S.c2d, Fin = x0.F, T = x0.T, sf = S.sf, st = S.stride, cout = S.cout;
    Act2 z;                                                   // ELU(x) with one zero row above and below
    z.C = C; z.F = Fin; z.T = T; z.halo = 1;
    z.buf = cx.alloc<float>((size_t)B * (Fin + 2) * C * T);
    const ConvGeom g = conv_geom(phases[0], T);               // time axis: Tout = T * st, count_T = (T + 1) * st
    int f_r = sf / 2;
    const int f_l = sf - f_r;
    if (last && f_r > 0) f_r -= 1;                            // last_out_padding [(0, 1), (0, 0)] (seanet_decoder.py:279,316)
    const int Fout = (Fin + 1) * sf - f_l - f_r;
    Act2 o;
    o.C = cout; o.F = Fout; o.T = g.Tout; o.halo = out_halo;
    const long long orow = (long long)cout * g.Tout;
    o.buf = cx.alloc<float>((size_t)B * (Fout + 2 * out_halo) * orow);
    fc::ConvLaunch c0;
    c0.M = phases[0].M; c0.BM = phases[0].BM; c0.BN = phases[0].BN; c0.Tout = T + 1;
    const int nblk = fc::conv_nblk(c0);
    const long long part_row = (long long)(Fin + 1) * nblk;
    const int gnblk = fc::gconvtr2d_nblk(T, st, Fin, sf);     // partial slots of the grouped direct kernel (same buffer)
    const bool has_norm = S.has_norm;
    const int t_trimL = S.causal ? 0 : st - st / 2;           // unpad2d: causal (trim_right_ratio 1) trims the time axis on the right only (conv.py:427-431)
    double* partials = has_norm ? cx.alloc<double>((size_t)B * std::max<long long>(sf * part_row, gnblk) * 2) : nullptr;
    o.aff = has_norm ? cx.alloc<float>((size_t)B * cout * 2) : nullptr;
    o.normed = has_norm;
    const double fl = 2.0 * B * (Fin + 1) * (double)phases[0].M * 2 * C * 2 * (T + 1) * sf;
    const double by = 4.0 * B * ((double)C * Fin * T * (x1 ? 2 : 1) + (double)cout * Fout * g.Tout);
    cx.conv_flops += fl; cx.conv_bytes += by;
    cx.launches += sf + 3 + (out_halo ? 1 : 0) - (halo_fuse_on() ? 1 : 0) - ((phases[0].w_group && halo_fuse_on() && out_halo && Fout > out_halo) ? 1 : 0);
    cx.conv_launches += sf;
    if (cx.dry || cx.err) return o;
    hipError_t er = fc::launch_combine2d(x0.buf, x0.aff, x0.halo, x1 ? x1->buf : nullptr, x1 ? x1->aff : nullptr, x1 ? x1->halo : 0, 1,
                                         e->arch.elu_alpha, B, Fin, C, T, z.buf, 1, cx.st, halo_fuse_on() ? 2 : 0);      // 2: its zero rows too
    if (er == hipSuccess && !halo_fuse_on()) er = fc::launch_halo_rows(z.buf, B, Fin, 1, C, T, 1, cx.st);
    if (phases[0].w_group) {          // grouped (2 in / 1 out channel per group): one direct launch over the untrimmed output
        double* gpart = partials;
        if (er == hipSuccess) {
            int cls = 0;
            if (e->profiling) {
                char nm[64];
                snprintf(nm, sizeof(nm), "gconvtr2d_kernel<%d>", st);
                cls = e->prof_class(nm);
            }
            ProfSpan sp(e, cx, cls, 2.0 * B * (double)cout * 8 * (Fin + 1) * sf * (T + 1) * st, by);
            er = fc::launch_gconvtr2d(z.buf, phases[0].w_group, phases[0].w_plain, o.buf + (long long)out_halo * orow, gpart, B, C, cout, Fin, T, sf, st,
                                      f_l, Fout, t_trimL, g.Tout, (long long)(Fout + 2 * out_halo) * orow, cx.st,
                                      (halo_fuse_on() && Fout > out_halo) ? out_halo : 0);
        }
        if (er == hipSuccess && has_norm)
            er = fc::launch_gn_finalize(gpart, gnblk, (double)cout * (Fin + 1) * sf * g.count_T, phases[0].gamma, phases[0].beta, cout, e->arch.gn_eps,
                                        B, o.aff, cx.st);
        if (er == hipSuccess && out_halo && !(halo_fuse_on() && Fout > out_halo)) er = fc::launch_halo_rows(o.buf, B, Fout, out_halo, cout, g.Tout, 0, cx.st);
        if (er != hipSuccess) { cx.err = 1; g_err = "grouped 2-D transposed conv launch failed (" + S.prefix + "): " + hipGetErrorString(er); }
        return o;
    }
    for (int p = 0; p < sf && er == hipSuccess; ++p) {
        const ConvLayer& L = phases[p];
        fc::ConvLaunch c;
        c.s0.ptr = z.buf; c.s0.used = 1;
        c.wt = L.wt; c.bias = L.bias; c.koff = L.koff;
        c.B = B * (Fin + 1); c.Cin = L.cin; c.Tin = T; c.M = L.M;
        c.k = L.gk; c.stride = L.gstride; c.dil = 1; c.padL = g.padL; c.padR = g.padR; c.pad_zero = 1;
        c.BM = L.BM; c.BN = L.BN; c.CC = L.CC; c.nchunk = L.nchunk; c.row = L.row ? 1 : 0;
        c.Tout = T + 1; c.up_r = st; c.trimL = t_trimL; c.Tfinal = g.Tout;
        c.Fo = Fin + 1; c.affC = C;
        c.in_sB0 = (long long)(Fin + 2) * C * T; c.in_sB1 = (long long)C * T;
        c.out = o.buf + ((long long)out_halo + p - f_l) * orow;      // rows outside [0, Fout) are never stored (store range below)
        c.out_sB = (long long)(Fout + 2 * out_halo) * orow; c.out_sF = (long long)sf * orow; c.out_sM = g.Tout; c.out_sT = 1;
        int lo = f_l - p;                                            // q * sf + p - f_l >= 0
        lo = lo <= 0 ? 0 : (lo + sf - 1) / sf;
        int hi = (Fout - 1 + f_l - p) >= 0 ? (Fout - 1 + f_l - p) / sf + 1 : 0;   // q * sf + p - f_l <= Fout - 1
        if (hi > Fin + 1) hi = Fin + 1;
        c.store_lo = lo; c.store_hi = hi;
        c.partials = has_norm ? partials + (long long)p * part_row * 2 : nullptr;
        c.part_sB0 = (long long)sf * part_row;
        int cls = 0;
        if (e->profiling) {
            int mode = 0, nu = 0, row = 0;
            fc::conv_variant(c, &mode, &nu, &row);
            char nm[80];
            snprintf(nm, sizeof(nm), "conv_mfma_kernel<%d, %d, %d, %d, %d, %d, %s, %s>", L.BM, L.BN, L.BM >= 128 ? 2 : 1, L.BM >= 128 ? 2 : 4, mode, nu,
                     (row & 1) ? "true" : "false", (row & 2) ? "true" : "false");
            if (fc::conv_cout1_ok(c))
                fc::conv_fewout_name(c, nm, sizeof(nm));
            cls = e->prof_class(nm);
        }
        ProfSpan sp(e, cx, cls, fl / sf, by / sf);
        er = fc::launch_conv(c, cx.st);
    }
    const ConvLayer& L0 = phases[0];
    if (er == hipSuccess && has_norm)
        er = fc::launch_gn_finalize(partials, (int)(sf * part_row), (double)cout * (Fin + 1) * sf * g.count_T, L0.gamma, L0.beta, cout, e->arch.gn_eps,
                                    B, o.aff, cx.st);
    if (er == hipSuccess && out_halo) er = fc::launch_halo_rows(o.buf, B, Fout, out_halo, cout, g.Tout, 0, cx.st);
    if (er != hipSuccess) { cx.err = 1; g_err = "2-D transposed conv launch failed (" + S.prefix + "): " + hipGetErrorString(er); }
    return o;
}

int stft_frames(const fc_engine* e, int T) { return 1 + T / e->arch.stft_hop; }

// fc_debug_freq_features: one-shot capture (mode 1) or override (mode 2) of the STFT-domain feature tensor of the next encode
struct FeatHook { float* buf = nullptr; size_t cap = 0; int mode = 0; };
thread_local FeatHook g_feat_hook;

// FreqCodec._encode_frame (codec_freq.py:330-392, mag_phase) + SEANetEncoder2d.forward: wav -> last conv (raw + affine) at Tf frames
Act run_encoder_2d(fc_engine* e, Ctx& cx, const float* wav, int T, const float* scale) {
    const fc_arch& a = e->arch;
    const int B = cx.B, hop = a.stft_hop, F = a.n_fft / 2 + 1, Tp = stft_frames(e, T), taps = ceil_div_i(a.n_fft, hop), Mp = Tp + taps - 1;
    float* xp = cx.alloc<float>((size_t)B * hop * Mp);
    cx.launches += 3;
    if (!cx.dry && !cx.err) {
        if (T <= a.n_fft / 2) { cx.err = 1; g_err = "utterance shorter than n_fft/2 + 1 samples (torch.stft reflect padding needs more)"; }
        else if (fc::launch_polyphase_in(wav, a.audio_normalize ? scale : nullptr, B, T, hop, a.n_fft, Mp, xp, cx.st) != hipSuccess) {
            cx.err = 1; g_err = "polyphase launch failed";
        }
    }
    fc::Src sx; sx.ptr = xp; sx.used = 1;
    Act spec = run_conv(e, cx, e->stft, sx, fc::Src(), 0, Mp);                     // [B][2F][Tp]: re rows, then im rows
    Act2 feats;
    feats.C = a.input_channels; feats.F = F; feats.T = Tp; feats.halo = e->halo2;
    feats.buf = cx.alloc<float>((size_t)B * (F + 2 * e->halo2) * feats.C * Tp);
    if (!cx.dry && !cx.err) {
        hipError_t er = fc::launch_stft_feats(spec.raw, B, F, Tp, (long long)2 * F * Tp, e->halo2, feats.C, feats.buf, cx.st);
        // test hooks (fc_debug_freq_features): hand the features out / take them from the caller, in the reference's [B][C][F][Tp]
        if (er == hipSuccess && g_feat_hook.buf && g_feat_hook.mode != 0) {
            const size_t need = (size_t)B * feats.C * F * Tp * sizeof(float);
            if (need > g_feat_hook.cap) { cx.err = 1; g_err = "fc_debug_freq_features: buffer too small"; }
            else er = fc::launch_feats_relayout(feats.buf, g_feat_hook.buf, B, feats.C, F, Tp, e->halo2, g_feat_hook.mode == 1, cx.st);
            g_feat_hook.mode = 0;            // one shot
        }
        if (er == hipSuccess) er = fc::launch_halo_rows(feats.buf, B, F, e->halo2, feats.C, Tp, 0, cx.st);
        if (er != hipSuccess) { cx.err = 1; g_err = std::string("stft feature launch failed: ") + hipGetErrorString(er); }
    }
    Act2 x = run_conv2d(e, cx, e->enc2_first, feats, nullptr, 0, e->halo2);
    for (size_t si = 0; si < e->enc_stages.size(); ++si) {
        auto& S = e->enc_stages[si];
        Act2 sc, b3;
        run_resblocks2d(e, cx, S, x, nullptr, &sc, &b3);
        x = run_conv2d(e, cx, S.resample, sc, &b3, 1, si + 1 == e->enc_stages.size() ? 0 : e->halo2);
    }
    if (!cx.dry && !cx.err && x.F != 1) { cx.err = 1; g_err = "the 2-D encoder must reduce the frequency axis to one bin (n_fft / ratios mismatch)"; }
    Act x1;                                                                        // ReshapeModule: [B][1][C][T] is [B][C][T]
    x1.raw = x.buf; x1.aff = x.aff; x1.C = x.C; x1.T = x.T; x1.normed = x.normed;
    if (e->enc_lstm.H) {
        Act y = run_lstm(e, cx, e->enc_lstm, x1, x1.T);
        if (a.lstm_skip) return run_conv(e, cx, e->enc_last, src_of(y), src_of(x1), 1, x1.T);
        return run_conv(e, cx, e->enc_last, src_of(y), fc::Src(), 1, x1.T);
    }
    return run_conv(e, cx, e->enc_last, src_of(x1), fc::Src(), 1, x1.T);
}

// SEANetDecoder2d.forward + FreqCodec._decode_frame (codec_freq.py:409-448, mag_phase): z [B][D][Tf] -> wav [B][out_len]
void run_decoder_2d(fc_engine* e, Ctx& cx, const float* z_bdt, int Tf, const float* scale, int out_len, float* wav) {
    const fc_arch& a = e->arch;
    const int B = cx.B, hop = a.stft_hop, F = a.n_fft / 2 + 1, taps = ceil_div_i(a.n_fft, hop);
    fc::Src s; s.ptr = z_bdt; s.used = 1;
    Act x = run_conv(e, cx, e->dec_first, s, fc::Src(), 0, Tf);
    Act2 a0, a1;
    bool has1 = false;
    auto as2 = [](const Act& t) { Act2 r; r.buf = t.raw; r.aff = t.aff; r.C = t.C; r.F = 1; r.T = t.T; r.halo = 0; r.normed = t.normed; return r; };
    a0 = as2(x);
    if (e->dec_lstm.H) {
        Act y = run_lstm(e, cx, e->dec_lstm, x, x.T);
        a0 = as2(y);
        if (a.lstm_skip) { a1 = as2(x); has1 = true; }
    }
    for (size_t si = 0; si < e->dec_stages.size(); ++si) {
        auto& S = e->dec_stages[si];
        Act2 up = run_convtr2d(e, cx, S.resample, e->dec_up_phases[si], a0, has1 ? &a1 : nullptr, si + 1 == e->dec_stages.size(), e->halo2);
        Act2 sc, b3;
        run_resblocks2d(e, cx, S, up, nullptr, &sc, &b3);
        a0 = sc; a1 = b3; has1 = true;
    }
    Act2 last = run_conv2d(e, cx, e->dec2_last, a0, &a1, 1, 0);                     // [B][F][3][Tp2] raw + GroupNorm(1, 3) affine
    const int Tp2 = last.T, Mp2 = Tp2 + taps - 1;
    float* spec = cx.alloc<float>((size_t)B * 2 * F * Tp2);
    cx.launches += 2;
    if (!cx.dry && !cx.err) {
        if (last.F != F) { cx.err = 1; g_err = "internal: decoder frequency rows != n_fft / 2 + 1"; }
        else if (out_len > hop * (Tp2 - 1)) { cx.err = 1; g_err = "out_len exceeds the inverse STFT's length stft_hop * (frames - 1)"; }
        else if (fc::launch_spec_from_dec(last.buf, last.aff, B, F, Tp2, 0, a.input_channels, spec, cx.st) != hipSuccess) { cx.err = 1; g_err = "spectrum launch failed"; }
    }
    fc::Src ss; ss.ptr = spec; ss.used = 1;
    Act yp = run_conv(e, cx, e->istft, ss, fc::Src(), 0, Tp2);                      // [B][hop][Mp2]
    if (!cx.dry && !cx.err) {
        if (yp.T != Mp2) { cx.err = 1; g_err = "internal: inverse-STFT GEMM length"; }
        else if (fc::launch_istft_finish(yp.raw, e->win2, B, hop, a.n_fft, Mp2, Tp2, scale, out_len, wav, cx.st) != hipSuccess) {
            cx.err = 1; g_err = "istft finish launch failed";
        }
    }
}

int total_hop(const fc_engine* e) {
    int h = e->arch.model_type == 1 ? e->arch.stft_hop : 1;
    for (int i = 0; i < e->arch.n_ratios; ++i) h *= e->arch.ratios[i];
    return h;
}

int decoded_samples(const fc_engine* e, int Tf) {
    if (e->arch.model_type != 1) return Tf * total_hop(e);
    int tp = Tf;
    for (int i = 0; i < e->arch.n_ratios; ++i) tp *= e->arch.ratios[i];
    return e->arch.stft_hop * (tp - 1);                                   // torch.istft, center = True
}

int frames_for(const fc_engine* e, int T) {
    if (e->arch.model_type == 1) T = stft_frames(e, T);
    for (int i = e->arch.n_ratios - 1; i >= 0; --i) T = ceil_div_i(T, e->arch.ratios[i]);
    return T;
}

// ---- composite paths (each usable in dry mode for workspace sizing / work accounting) ---------------
int do_encode(fc_engine* e, Ctx& cx, const float* wav, int T, int n_q, int64_t* codes, float* quantized,
              float* sub_quants, float* scale, float* enc_out, float** quant_bdt_out) {
    const int B = cx.B, D = e->arch.dimension, Tf = frames_for(e, T);
    float* sc = nullptr;
    if (e->arch.audio_normalize) {
        sc = scale ? scale : cx.alloc<float>(B);
        cx.launches++;
        if (!cx.dry && !cx.err) {
            if (fc::launch_volume(wav, B, e->audio_ch(), T, sc, cx.st) != hipSuccess) return fail("volume kernel launch failed");
        }
    }
    Act last = e->arch.model_type == 1 ? run_encoder_2d(e, cx, wav, T, sc) : run_encoder(e, cx, wav, T, sc);
    const int Dc = e->cdim();
    const bool ranged = e->arch.codec_range > 0.f;
    float* emb = enc_out ? enc_out : ((e->q_proj || ranged) && !enc_out ? nullptr : cx.alloc<float>((size_t)B * Tf * D));
    // quantiser input rows [B*Tf][Dc]: the encoder output itself, or input_proj(...) / tanh(...) * range of it (costume_quantizer.py:84-87)
    Act pj;
    if (e->q_proj) pj = run_conv(e, cx, e->q_in, src_of(last), fc::Src(), 0, Tf);
    float* xq = (e->q_proj || ranged) ? cx.alloc<float>((size_t)B * Tf * Dc) : emb;
    float* quant_c = e->q_proj ? cx.alloc<float>((size_t)B * Tf * Dc) : quantized;       // quantised rows in codebook space
    float* qbdt_c = cx.alloc<float>((size_t)B * Dc * Tf);
    // quantizer_conf.q0_ds_ratio > 1 (ddp_core_vq.py:396-404): stage 0 of frame t quantises frame q0_source_frame(t, Tf) (kernels.h)
    const bool q0 = e->arch.q0_ds_ratio > 1;
    int* q0map = q0 ? cx.alloc<int>((size_t)B * Tf) : nullptr;
    cx.launches += 2 + (ranged ? 1 : 0) + (q0 ? 1 : 0);
    cx.rvq_flops += 2.0 * B * Tf * (double)n_q * e->arch.codebook_size * Dc;
    if (!cx.dry && !cx.err) {
        if (last.T != Tf) return fail("internal: frame count mismatch");
        // encoder output permuted to [B,Tf,D] (seanet_encoder.py:175) with the last GroupNorm applied
        if (emb && fc::launch_combine(src_of(last), fc::Src(), 0, 1.f, nullptr, B, D, Tf, Tf, emb, (long long)Tf * D, 1, D, cx.st) != hipSuccess)
            return fail("combine launch failed");
        if (e->q_proj) {
            if (fc::launch_combine(src_of(pj), fc::Src(), 0, 1.f, nullptr, B, Dc, Tf, Tf, xq, (long long)Tf * Dc, 1, Dc, cx.st) != hipSuccess)
                return fail("combine launch failed");
        } else if (ranged) {
            if (fc::launch_combine(src_of(last), fc::Src(), 0, 1.f, nullptr, B, D, Tf, Tf, xq, (long long)Tf * D, 1, D, cx.st) != hipSuccess)
                return fail("combine launch failed");
        }
        if (ranged && fc::launch_tanh_range(xq, (size_t)B * Tf * Dc, e->arch.codec_range, cx.st) != hipSuccess) return fail("tanh launch failed");
        if (q0) {
            if (Tf < 2) return fail("quantizer_conf.q0_ds_ratio > 1 needs at least 2 frames (the reference's F.interpolate(size=[Tf // 2]) raises on 0)");
            if (fc::launch_q0_map(q0map, B, Tf, cx.st) != hipSuccess) return fail("q0 map launch failed");
        }
        ProfSpan sp(e, cx, e->profiling ? e->prof_class(kRvqClass) : 0, 2.0 * B * Tf * (double)n_q * e->arch.codebook_size * Dc, 0.0);
        if (fc::launch_rvq_encode(xq, B * Tf, Dc, e->arch.codebook_size, n_q, e->cb, e->cb_frag, e->enorm, codes, quant_c, qbdt_c,
                                  sub_quants, Tf, cx.st, q0map) != hipSuccess)
            return fail("rvq launch failed (codebook size must be a multiple of 64, dim in {16,32,64,128,256,512})");
    }
    float* qbdt = qbdt_c;
    if (e->q_proj) {       // output_proj (costume_quantizer.py:92-94): decoder input [B][D][Tf] and the returned embeddings [B][Tf][D]
        fc::Src qs; qs.ptr = qbdt_c; qs.used = 1;
        Act qo = run_conv(e, cx, e->q_out, qs, fc::Src(), 0, Tf);
        qbdt = qo.raw;
        cx.launches++;
        if (!cx.dry && !cx.err && quantized &&
            fc::launch_combine(src_of(qo), fc::Src(), 0, 1.f, nullptr, B, D, Tf, Tf, quantized, (long long)Tf * D, 1, D, cx.st) != hipSuccess)
            return fail("combine launch failed");
    }
    if (quant_bdt_out) *quant_bdt_out = qbdt;
    return cx.err;
}

int do_decode(fc_engine* e, Ctx& cx, const float* z_bdt, int Tf, const float* scale, int out_len, float* wav) {
    if (e->arch.model_type == 1) {
        run_decoder_2d(e, cx, z_bdt, Tf, scale, out_len, wav);
        return cx.err;
    }
    Act last = run_decoder(e, cx, z_bdt, Tf);
    cx.launches++;
    if (!cx.dry && !cx.err) {
        if (out_len > last.T) return fail("out_len exceeds Tf*hop");
        // final GroupNorm apply (decoder.model.N.conv.norm has C = audio channels), x scale (codec_basic.py:406-407), trim (:711)
        const int C = e->audio_ch();
        if (fc::launch_combine(src_of(last), fc::Src(), 0, 1.f, scale, cx.B, C, last.T, out_len, wav, (long long)C * out_len, out_len, 1, cx.st) != hipSuccess)
            return fail("combine launch failed");
    }
    return cx.err;
}

// Deferred device-side failures of EARLIER calls (kernels cannot return a status): reported once, then cleared.
int consume_status(fc_engine* e) {
    if (!e->status_host) return 0;
    if (e->status_host[FC_STATUS_LSTM_TIMEOUT]) {
        e->status_host[FC_STATUS_LSTM_TIMEOUT] = 0;
        e->lstm_persist_ok = false;
        return fail("a previous call's persistent LSTM kernel timed out at its grid barrier (its workgroups were not all co-resident: "
                    "GPU shared with another process / stream, or CUs masked); that call's outputs were NaN-poisoned and are invalid. "
                    "This engine now uses the per-step LSTM launches (same results); set FC_LSTM_PERSIST=0 for shared-GPU deployments");
    }
    if (e->status_host[FC_STATUS_BAD_CODE]) {
        e->status_host[FC_STATUS_BAD_CODE] = 0;
        return fail("a previous decode call was given code indices outside [0, codebook_size) (the reference's F.embedding raises, "
                    "ddp_core_vq.py:191); they were clamped, the decoded audio of that call is not meaningful");
    }
    return 0;
}

int check_ready(fc_engine* e) {
    if (!e) return fail("null engine");
    if (!e->finalized) return fail("engine not finalized");
    return consume_status(e);
}

Ctx make_ctx(int B, void* ws, size_t ws_bytes, void* stream) {
    Ctx cx;
    cx.B = B; cx.st = (hipStream_t)stream; cx.base = (char*)ws; cx.cap = ws_bytes;
    return cx;
}

}  // namespace

// ====================================================================================================
extern "C" {

int fc_abi_version(void) { return FC_ABI_VERSION; }
const char* fc_last_error(void) { return g_err.c_str(); }
// the other translation units of the library (laura.hip) report through the same thread-local message
void fc_set_last_error_(const char* msg) { g_err = msg ? msg : ""; }

int fc_engine_create(const fc_arch* arch, int device, fc_engine** out) {
    if (!arch || !out) return fail("null argument");
    if (arch->abi_version != FC_ABI_VERSION) return fail("fc_arch.abi_version mismatch");
    if (arch->n_residual_layers < 1 || arch->n_residual_layers > 8 || arch->dilation_base < 1)
        return fail("fc_arch.n_residual_layers must be in 1..8 and dilation_base >= 1");
    if (arch->norm_type < 0 || arch->norm_type > 2) return fail("fc_arch.norm_type must be 0 (GroupNorm), 1 (weight_norm) or 2 (none)");
    if (arch->norm_type == 0 && arch->causal) return fail("GroupNorm convs cannot be causal (the reference refuses it too, conv.py:46-47)");
    if (arch->n_ratios < 1 || arch->n_ratios > FC_MAX_RATIOS) return fail("n_ratios out of range");
    if (arch->compress < 1 || arch->n_filters < 2 || (arch->n_filters % arch->compress) != 0) return fail("bad n_filters/compress");
    if (arch->n_filters % 2) return fail("n_filters must be even");
    const int D = arch->dimension;
    if (!(D == 16 || D == 32 || D == 64 || D == 128 || D == 256 || D == 512)) return fail("dimension must be one of 16/32/64/128/256/512");
    if (arch->codec_dim < 0 || arch->codec_range < 0.f) return fail("fc_arch.codec_dim / codec_range must be >= 0");
    if (arch->codec_dim > 0) {
        const int Dc = arch->codec_dim;
        if (!(Dc == 16 || Dc == 32 || Dc == 64 || Dc == 128 || Dc == 256 || Dc == 512)) return fail("codec_dim must be one of 16/32/64/128/256/512");
    }
    if (arch->codebook_size % 64 || (arch->codebook_size > 128 && arch->codebook_size % 128))
        return fail("codebook_size must be a multiple of 64, and of 128 above 128 (8 waves x 16-code tiles)");
    if (arch->lstm_layers > 0 && ((arch->n_filters << arch->n_ratios) % 16)) return fail("LSTM width must be a multiple of 16");
    if (arch->lstm_layers > FC_LSTM_MAX_LAYERS) return fail("too many LSTM layers");
    for (int i = 0; i < arch->n_ratios; ++i)
        if (arch->ratios[i] < 1) return fail("ratios must be >= 1");
    if (arch->model_type != 0 && arch->model_type != 1) return fail("fc_arch.model_type must be 0 (encodec) or 1 (freq_codec, mag_phase)");
    if (arch->model_type == 0 && (arch->input_channels < 0 || arch->input_channels > 2))
        return fail("encodec: input_channels must be 1 (0 reads as 1) or 2 (stereo; the reference asserts channels <= 2, codec_basic.py:344)");
    if (arch->model_type == 1) {
        if (arch->norm_type != 0 && arch->norm_type != 1) return fail("freq_codec: norm must be time_group_norm or weight_norm");
        if (arch->input_channels != 3 && arch->input_channels != 2)
            return fail("freq_codec: input_channels must be 3 (codec_domain mag_phase: log-magnitude, phase re, phase im) or 2 (mag_angle: log-magnitude, angle)");
        if (arch->n_fft < 64 || (arch->n_fft & (arch->n_fft - 1)) || arch->stft_hop < 1 || arch->stft_hop > arch->n_fft)
            return fail("freq_codec: n_fft must be a power of two >= 64 and 1 <= stft_hop <= n_fft");
        if (arch->stft_hop % 2) return fail("freq_codec: stft_hop must be even");
        if (arch->dilation_base != 1 && arch->n_residual_layers > 1) return fail("freq_codec: dilated residual stacks are not built for the 2-D nets");
        int f = arch->n_fft / 2 + 1;
        for (int i = arch->n_ratios - 1; i >= 0; --i) {
            const int fr = arch->ratios_f[i];
            if (fr < 1) return fail("freq_codec: ratios_f must be >= 1");
            const int tot = (2 * fr - 1) - (fr - 1);
            f = (f + tot - 2 * fr) / fr + 1;
        }
        if (f != 1) return fail("freq_codec: the frequency ratios must reduce n_fft / 2 + 1 bins to exactly one");
    }
    fc_engine* e = new fc_engine();
    e->arch = *arch;
    e->device = device;
    build_plan(e);
    for (const auto& kv : e->by_prefix)
        if (kv.second->unsupported) {
            const std::string msg = "stride " + std::to_string(kv.second->gstride) + " / kernel " + std::to_string(kv.second->gk) + " of " + kv.first +
                                    " exceeds the conv kernel's slab (ratios above 16 are not built)";
            delete e;
            return fail(msg);
        }
    if (arch->model_type == 1) {          // torch.nn.Conv2d / ConvTranspose2d constraints on `groups` (conv_group_ratio > 0)
        std::vector<const ConvLayer*> ls;
        for (auto* st : {&e->enc_stages, &e->dec_stages})
            for (auto& S : *st) {
                for (auto& R : S.res) { ls.push_back(&R.shortcut); ls.push_back(&R.block1); ls.push_back(&R.block3); }
                ls.push_back(&S.resample);
            }
        for (const ConvLayer* L : ls)
            if (L->groups < 1 || L->c2d % L->groups || L->cout % L->groups) {
                const std::string msg = "freq_codec: conv_group_ratio gives " + std::to_string(L->groups) + " groups for " + L->prefix + " (" +
                                        std::to_string(L->c2d) + " -> " + std::to_string(L->cout) + " channels)";
                delete e;
                return fail(msg);
            }
    }
    *out = e;
    return 0;
}

int fc_engine_profile(fc_engine* e, int enable) {
    if (!e) return fail("null engine");
    e->profiling = enable != 0;
    return 0;
}

int fc_engine_profile_read(fc_engine* e, fc_prof* out) {
    if (!e || !out) return fail("null argument");
    for (int i = 0; i < FC_PROF_CLASSES; ++i) {
        memset(&out[i], 0, sizeof(fc_prof));
        if (i < (int)e->prof_names.size()) strncpy(out[i].kernel, e->prof_names[i].c_str(), sizeof(out[i].kernel) - 1);
    }
    if (!e->spans.empty()) HIP_TRY(hipEventSynchronize(e->spans.back().b));
    for (auto& s : e->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) continue;
        out[s.cls].total_ms += ms; out[s.cls].flops += s.flops; out[s.cls].bytes += s.bytes; out[s.cls].launches += 1;
    }
    e->spans.clear();
    e->events_used = 0;
    return 0;
}

void fc_engine_destroy(fc_engine* e) {
    if (!e) return;
    for (hipEvent_t ev : e->event_pool) (void)hipEventDestroy(ev);
    for (void* p : e->dev_allocs) (void)hipFree(p);
    if (e->status_host) (void)hipHostFree((void*)e->status_host);
    delete e;
}

int fc_engine_num_weights(const fc_engine* e) { return e ? (int)e->expected.size() : 0; }

int fc_engine_weight_info(const fc_engine* e, int i, const char** name, int64_t* dims) {
    if (!e || i < 0 || i >= (int)e->expected.size()) return -1;
    if (name) *name = e->expected[i].first.c_str();
    const auto& d = e->expected[i].second;
    if (dims) for (size_t j = 0; j < d.size(); ++j) dims[j] = d[j];
    return (int)d.size();
}

int fc_engine_set_weight(fc_engine* e, const char* name, const float* host, const int64_t* dims, int ndim) {
    if (!e || !name || !host || !dims) return fail("null argument");
    if (e->finalized) return fail("engine already finalized");
    for (auto& ex : e->expected) {
        if (ex.first != name) continue;
        if ((int)ex.second.size() != ndim) return fail(std::string("rank mismatch for ") + name);
        size_t n = 1;
        for (int j = 0; j < ndim; ++j) {
            if (ex.second[j] != dims[j]) return fail(std::string("shape mismatch for ") + name);
            n *= (size_t)dims[j];
        }
        HostTensor& t = e->host[name];
        t.dims.assign(dims, dims + ndim);
        t.data.assign(host, host + n);
        t.set = true;
        return 0;
    }
    g_err = std::string("tensor not part of the hot path: ") + name;
    return 2;
}

int fc_engine_finalize(fc_engine* e) {
    if (!e) return fail("null engine");
    if (e->finalized) return 0;
    for (auto& ex : e->expected)
        if (!e->host.count(ex.first) || !e->host[ex.first].set) return fail("checkpoint is missing tensor " + ex.first);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("no HIP device visible: the FunCodec MI355X engine has no CPU fallback");
    HIP_TRY(hipSetDevice(e->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, e->device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    if (e->arch.model_type == 1) {
        if (pack_conv2d(e, e->enc2_first) || pack_conv2d(e, e->dec2_last)) return 1;
        for (auto& S : e->enc_stages) {
            for (auto& R : S.res)
                if (pack_conv2d(e, R.shortcut) || pack_conv2d(e, R.block1) || pack_conv2d(e, R.block3)) return 1;
            if (pack_conv2d(e, S.resample)) return 1;
        }
        for (size_t si = 0; si < e->dec_stages.size(); ++si) {
            auto& S = e->dec_stages[si];
            for (auto& R : S.res)
                if (pack_conv2d(e, R.shortcut) || pack_conv2d(e, R.block1) || pack_conv2d(e, R.block3)) return 1;
            if (pack_convtr2d(e, S.resample, e->dec_up_phases[si])) return 1;
        }
        if (pack_conv(e, e->enc_last) || pack_conv(e, e->dec_first)) return 1;
        if (pack_dft(e)) return 1;
    }
    std::vector<ConvLayer*> convs = {&e->enc_first, &e->enc_last, &e->dec_first, &e->dec_last};
    if (e->arch.model_type == 1) convs.clear();
    for (auto* st : {&e->enc_stages, &e->dec_stages})
        for (auto& S : *st) {
            if (e->arch.model_type == 1) break;
            for (auto& R : S.res) { convs.push_back(&R.shortcut); convs.push_back(&R.block1); convs.push_back(&R.block3); }
            convs.push_back(&S.resample);
        }
    // fused res-block heads read the host copies of the two convs' weights: pack them before pack_conv drops nothing (host map is
    // still alive here)
    for (auto* stg : {&e->enc_stages, &e->dec_stages})
        for (auto& S : *stg)
            for (auto& R : S.res)
                if (e->arch.model_type == 0 && pack_reshead(e, R)) return 1;
    for (ConvLayer* L : convs)
        if (pack_conv(e, *L)) return 1;
    if (pack_lstm(e, e->enc_lstm)) return 1;
    if (pack_lstm(e, e->dec_lstm)) return 1;
    // codebooks + |e|^2 (EuclideanCodebook.quantize ddp_core_vq.py:185: embed.pow(2).sum(0)); sequential d, squares rounded
    const auto& E = e->host["quantizer.rq.model.embed"].data;
    const int nq = e->arch.num_quantizers, K = e->arch.codebook_size, D = e->cdim();
    if (e->q_proj) {       // Linear weights [out][in] are the k = 1 GEMM's [M][cin][1]
        if (pack_gemm(e, e->q_in, e->host["quantizer.input_proj.weight"].data, e->host["quantizer.input_proj.bias"].data)) return 1;
        if (pack_gemm(e, e->q_out, e->host["quantizer.output_proj.weight"].data, e->host["quantizer.output_proj.bias"].data)) return 1;
    }
    std::vector<float> en((size_t)nq * K);
    for (size_t r = 0; r < (size_t)nq * K; ++r) {
        volatile float s = 0.f;
        for (int d = 0; d < D; ++d) {
            volatile float sq = E[r * D + d] * E[r * D + d];
            s = s + sq;
        }
        en[r] = s;
    }
    if (upload(e, E, &e->cb)) return 1;
    if (K % 16 == 0 && D % 16 == 0) {
        // fragment order: [stage][16-code tile][q = d/16][g = (d%16)/4][code in tile][d%4]: lane (g, code) of a wave reads
        // 16 contiguous bytes and a whole wave instruction reads 1 KiB contiguous (row-major rows give 64-byte pieces)
        std::vector<float> F((size_t)nq * K * D);
        for (int i = 0; i < nq; ++i)
            for (int c = 0; c < K; ++c)
                for (int d = 0; d < D; ++d) {
                    const size_t tile = (size_t)i * (K / 16) + c / 16;
                    F[((tile * (D / 16) + d / 16) * 4 + (d % 16) / 4) * 64 + (size_t)(c % 16) * 4 + d % 4] = E[((size_t)i * K + c) * D + d];
                }
        if (upload(e, F, &e->cb_frag)) return 1;
    }
    if (upload(e, en, &e->enorm)) return 1;
    {   // status words: pinned host memory the kernels write to directly (error paths only)
        void* hp = nullptr;
        HIP_TRY(hipHostMalloc(&hp, FC_STATUS_WORDS * sizeof(unsigned), hipHostMallocMapped));
        memset(hp, 0, FC_STATUS_WORDS * sizeof(unsigned));
        void* dp = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
        e->status_host = (volatile unsigned*)hp;
        e->status_dev = (unsigned*)dp;
    }
    e->host.clear();                                  // host copies of the checkpoint are dropped
    HIP_TRY(hipDeviceSynchronize());
    e->finalized = true;
    return 0;
}

int fc_engine_hop_length(const fc_engine* e) { return e ? total_hop(e) : 0; }
int fc_engine_frames(const fc_engine* e, int n_samples) { return e ? frames_for(e, n_samples) : 0; }
int fc_engine_decoded_samples(const fc_engine* e, int n_frames) { return e ? decoded_samples(e, n_frames) : 0; }

size_t fc_engine_workspace_bytes(const fc_engine* ce, int B, int T) {
    fc_engine* e = const_cast<fc_engine*>(ce);
    if (!e || B <= 0 || T <= 0) return 0;
    // tiling is only known after finalize(); plan with the same rule here
    const int Tf = frames_for(e, T), D = e->arch.dimension;
    Ctx cx; cx.B = B; cx.dry = true;
    float* q = nullptr;
    do_encode(e, cx, nullptr, T, e->arch.num_quantizers, nullptr, nullptr, nullptr, nullptr, nullptr, &q);
    cx.alloc<float>((size_t)B * Tf * D);           // quantized when the caller does not want it
    cx.alloc<float>((size_t)B * Tf * D);           // emb for decode_codes
    cx.alloc<float>((size_t)B * Tf * D);           // transposed copy for decode_emb
    do_decode(e, cx, nullptr, Tf, nullptr, decoded_samples(e, Tf), nullptr);
    return cx.off + 4096;
}

int fc_engine_work(const fc_engine* ce, int B, int T, int n_q, fc_work* out) {
    fc_engine* e = const_cast<fc_engine*>(ce);
    if (!e || !out) return fail("null argument");
    Ctx cx; cx.B = B; cx.dry = true;
    float* q = nullptr;
    do_encode(e, cx, nullptr, T, n_q, nullptr, nullptr, nullptr, nullptr, nullptr, &q);
    do_decode(e, cx, nullptr, frames_for(e, T), nullptr, T, nullptr);
    out->conv_flops = cx.conv_flops; out->conv_bytes = cx.conv_bytes;
    out->lstm_flops = cx.lstm_flops; out->rvq_flops = cx.rvq_flops;
    out->total_flops = cx.conv_flops + cx.lstm_flops + cx.rvq_flops;
    out->total_bytes = cx.conv_bytes;
    out->conv_launches = cx.conv_launches; out->total_launches = cx.launches;
    return 0;
}

namespace {
// fc_debug_freq_features is ONE SHOT: whatever the next fc_encode / fc_encode_decode call of this thread does -- a time-domain model, an early
// error, success -- the hook (a raw device pointer the caller may free afterwards) is disarmed when that call returns
struct FeatHookDisarm { ~FeatHookDisarm() { g_feat_hook = FeatHook(); } };
}  // namespace

int fc_encode(fc_engine* e, const float* wav, int B, int T, int n_q, int64_t* codes, float* quantized, float* sub_quants,
              float* scale, float* enc_out, void* workspace, size_t workspace_bytes, void* stream) {
    FeatHookDisarm disarm_feature_hook;
    if (check_ready(e)) return 1;
    if (!wav || !codes || B <= 0 || T <= 0) return fail("bad argument");
    if (n_q < 1 || n_q > e->arch.num_quantizers) return fail("n_q out of range");
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    return do_encode(e, cx, wav, T, n_q, codes, quantized, sub_quants, scale, enc_out, nullptr);
}

int fc_decode_emb(fc_engine* e, const float* emb, const float* scale, int B, int Tf, int out_len, float* wav,
                  void* workspace, size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    if (!emb || !wav || B <= 0 || Tf <= 0 || out_len <= 0) return fail("bad argument");
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    const int D = e->arch.dimension;
    float* z = cx.alloc<float>((size_t)B * D * Tf);
    if (cx.err) return 1;
    HIP_TRY(fc::launch_transpose_btd(emb, B, Tf, D, z, cx.st));
    return do_decode(e, cx, z, Tf, scale, out_len, wav);
}

int fc_decode_codes(fc_engine* e, const int64_t* codes, int B, int Tf, int n_q, int out_len, float* wav, float* emb_out,
                    void* workspace, size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    if (!codes || !wav || B <= 0 || Tf <= 0 || out_len <= 0) return fail("bad argument");
    if (n_q < 1 || n_q > e->arch.num_quantizers) return fail("n_q out of range");
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    const int D = e->arch.dimension, Dc = e->cdim();
    float* z = cx.alloc<float>((size_t)B * Dc * Tf);
    if (cx.err) return 1;
    HIP_TRY(fc::launch_rvq_decode(codes, B, Tf, n_q, Dc, e->arch.codebook_size, e->cb, e->q_proj ? nullptr : emb_out, z, e->status_dev, cx.st));
    if (e->q_proj) {       // CostumeQuantizer.decode (costume_quantizer.py:114-119): output_proj on the summed code vectors
        fc::Src qs; qs.ptr = z; qs.used = 1;
        Act qo = run_conv(e, cx, e->q_out, qs, fc::Src(), 0, Tf);
        if (cx.err) return 1;
        if (emb_out) HIP_TRY(fc::launch_combine(src_of(qo), fc::Src(), 0, 1.f, nullptr, B, D, Tf, Tf, emb_out, (long long)Tf * D, 1, D, cx.st));
        z = qo.raw;
    }
    return do_decode(e, cx, z, Tf, nullptr, out_len, wav);
}

int fc_encode_decode(fc_engine* e, const float* wav, int B, int T, int n_q, int use_scale, int64_t* codes, float* quantized,
                     float* sub_quants, float* scale, float* recon, void* workspace, size_t workspace_bytes, void* stream) {
    FeatHookDisarm disarm_feature_hook;
    if (check_ready(e)) return 1;
    if (!wav || !codes || !recon || B <= 0 || T <= 0) return fail("bad argument");
    if (n_q < 1 || n_q > e->arch.num_quantizers) return fail("n_q out of range");
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    float* sc = scale;
    if (e->arch.audio_normalize && !sc) sc = cx.alloc<float>(B);
    float* qbdt = nullptr;
    if (do_encode(e, cx, wav, T, n_q, codes, quantized, sub_quants, sc, nullptr, &qbdt)) return 1;
    const int Tf = frames_for(e, T);
    const int dec_len = decoded_samples(e, Tf);                     // recon is [B, min(T, decoded samples)] (the reference's recon[:, :, :T])
    return do_decode(e, cx, qbdt, Tf, (use_scale && e->arch.audio_normalize) ? sc : nullptr, T < dec_len ? T : dec_len, recon);
}

int fc_rvq_encode(fc_engine* e, const float* x, int N, int n_q, int64_t* codes, float* quantized, void* workspace,
                  size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    (void)workspace; (void)workspace_bytes;
    if (!x || !codes || N <= 0) return fail("bad argument");
    if (n_q < 1 || n_q > e->arch.num_quantizers) return fail("n_q out of range");
    int* q0map = nullptr;
    if (e->arch.q0_ds_ratio > 1) {      // the rows are ONE utterance of N frames; the stage-0 source-row table lives in the caller's workspace
        if (N < 2) return fail("quantizer_conf.q0_ds_ratio > 1 needs at least 2 frames");
        if (!workspace || workspace_bytes < (size_t)N * sizeof(int)) return fail("fc_rvq_encode with q0_ds_ratio > 1 needs a workspace of N * 4 bytes");
        q0map = (int*)workspace;
        HIP_TRY(fc::launch_q0_map(q0map, 1, N, (hipStream_t)stream));
    }
    HIP_TRY(fc::launch_rvq_encode(x, N, e->cdim(), e->arch.codebook_size, n_q, e->cb, e->cb_frag, e->enorm, codes, quantized,
                                  nullptr, nullptr, N, (hipStream_t)stream, q0map));
    return 0;
}

int fc_q0_source_frames(int Tf, int32_t* frames) {
    if (Tf < 2 || !frames) return fail("fc_q0_source_frames: Tf >= 2 and a host buffer of Tf entries");
    for (int t = 0; t < Tf; ++t) frames[t] = fc::q0_source_frame(t, Tf);
    return 0;
}

int fc_layer_out_len(const fc_engine* e, const char* prefix, int T) {
    if (!e || !prefix) return -1;
    auto it = e->by_prefix.find(prefix);
    if (it == e->by_prefix.end()) return -1;
    return conv_geom(*it->second, T).Tout;
}

int fc_layer_forward(fc_engine* e, const char* prefix, const float* x, int B, int T, int apply_elu, float* y,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    if (!prefix || !x || !y || B <= 0 || T <= 0) return fail("bad argument");
    auto it = e->by_prefix.find(prefix);
    if (it == e->by_prefix.end()) return fail(std::string("unknown layer ") + prefix);
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    fc::Src s; s.ptr = x; s.used = 1;
    Act o = run_conv(e, cx, *it->second, s, fc::Src(), apply_elu, T);
    if (cx.err) return 1;
    HIP_TRY(fc::launch_combine(src_of(o), fc::Src(), 0, 1.f, nullptr, B, o.C, o.T, o.T, y, (long long)o.C * o.T, o.T, 1, cx.st));
    return 0;
}

int fc_resblock_forward(fc_engine* e, const char* prefix, const float* x, int B, int T, float* y, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    if (!prefix || !x || !y || B <= 0 || T <= 0) return fail("bad argument");
    auto it = e->res_by_prefix.find(prefix);
    if (it == e->res_by_prefix.end()) return fail(std::string("unknown residual block ") + prefix);
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    fc_engine::Stage one;
    one.res.push_back(*it->second);
    fc::Src s; s.ptr = x; s.used = 1;
    Act sc, b3;
    run_resblocks(e, cx, one, s, fc::Src(), T, &sc, &b3);
    if (cx.err) return 1;
    HIP_TRY(fc::launch_combine(src_of(sc), src_of(b3), 0, 1.f, nullptr, B, sc.C, T, T, y, (long long)sc.C * T, T, 1, cx.st));
    return 0;
}

int fc_lstm_forward(fc_engine* e, const char* prefix, const float* x, int B, int T, float* y, void* workspace,
                    size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    if (!prefix || !x || !y || B <= 0 || T <= 0) return fail("bad argument");
    auto it = e->lstm_by_prefix.find(prefix);
    if (it == e->lstm_by_prefix.end()) return fail(std::string("unknown lstm ") + prefix);
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    Act in; in.raw = const_cast<float*>(x); in.C = it->second->H; in.T = T;
    Act o = run_lstm(e, cx, *it->second, in, T);
    if (cx.err) return 1;
    fc::Src s1;
    if (e->arch.lstm_skip) s1 = src_of(in);
    HIP_TRY(fc::launch_combine(src_of(o), s1, 0, 1.f, nullptr, B, o.C, T, T, y, (long long)o.C * T, T, 1, cx.st));
    return 0;
}

int fc_engine_status(fc_engine* e, unsigned* flags) {
    if (!e) return fail("null engine");
    unsigned f = 0;
    if (e->status_host) {
        if (e->status_host[FC_STATUS_LSTM_TIMEOUT]) f |= FC_STATUS_FLAG_LSTM_TIMEOUT;
        if (e->status_host[FC_STATUS_BAD_CODE]) f |= FC_STATUS_FLAG_BAD_CODE;
    }
    if (flags) *flags = f;
    if (!e->finalized) return 0;
    return consume_status(e);
}

int fc_overlap_add(const float* const* frames, const int* lens, int n_frames, int B, int frame0_len, int stride, int out_len,
                   float* out, void* stream) {
    if (!frames || !lens || !out || n_frames <= 0 || B <= 0 || frame0_len <= 0 || stride <= 0 || out_len <= 0) return fail("bad argument");
    HIP_TRY(fc::launch_overlap_add(frames, lens, n_frames, B, frame0_len, stride, out_len, out, (hipStream_t)stream));
    return 0;
}

size_t fc_codec_json_bound(int n_q, int len) { return (size_t)(n_q > 0 ? n_q : 0) * ((size_t)(len > 0 ? len : 0) * 22 + 4) + 8; }

int fc_format_codec_json(const int64_t* codes, int n_q, int B, int T, int b, int len, char* out, size_t cap, size_t* written) {
    if (!codes || !out || !written || n_q <= 0 || B <= 0 || T <= 0 || b < 0 || b >= B || len < 0 || len > T) return fail("bad argument");
    if (cap < fc_codec_json_bound(n_q, len)) return fail("output buffer too small (fc_codec_json_bound)");
    char* p = out;
    *p++ = '['; *p++ = '[';
    for (int q = 0; q < n_q; ++q) {
        if (q) { *p++ = ','; *p++ = ' '; }
        *p++ = '[';
        const int64_t* row = codes + ((size_t)q * B + b) * T;
        for (int t = 0; t < len; ++t) {
            if (t) { *p++ = ','; *p++ = ' '; }
            long long v = row[t];
            if (v < 0) { *p++ = '-'; v = -v; }
            char tmp[24];
            int n = 0;
            do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
            while (n) *p++ = tmp[--n];
        }
        *p++ = ']';
    }
    *p++ = ']'; *p++ = ']';
    *written = (size_t)(p - out);
    return 0;
}

int fc_write_wav_pcm16(const char* path, const float* wav, int n, int sample_rate, int rescale) {
    if (!path || !wav || n < 0 || sample_rate <= 0) return fail("bad argument");
    const float limit = 0.99f;
    float mx = 0.f;
    for (int i = 0; i < n; ++i) { const float a = fabsf(wav[i]); if (a > mx) mx = a; }
    // torch: wav * min(limit / mx, 1), else clamp(-limit, limit).  `limit / mx` with a 0-dim tensor mx is Tensor.__rtruediv__ =
    // mx.reciprocal() * limit, two fp32 roundings (1 ulp off limit / mx for about a quarter of all peaks)
    float mul = 1.f;
    if (rescale && mx > 0.f) { const float r = (1.0f / mx) * limit; mul = r < 1.f ? r : 1.f; }
    std::vector<int16_t> pcm((size_t)n);
    for (int i = 0; i < n; ++i) {
        float v = wav[i];
        if (rescale) v = v * mul;
        else v = v < -limit ? -limit : (v > limit ? limit : v);
        float s = nearbyintf(v * 32768.0f);            // torch.round: half to even
        s = s < -32768.f ? -32768.f : (s > 32767.f ? 32767.f : s);
        pcm[i] = (int16_t)s;
    }
    FILE* f = fopen(path, "wb");
    if (!f) return fail(std::string("cannot open ") + path);
    const uint32_t data_bytes = (uint32_t)n * 2u, riff = 36u + data_bytes, fmt_len = 16u, sr = (uint32_t)sample_rate, byte_rate = sr * 2u;
    const uint16_t pcm_tag = 1, ch = 1, align = 2, bits = 16;
    bool ok = fwrite("RIFF", 1, 4, f) == 4 && fwrite(&riff, 4, 1, f) == 1 && fwrite("WAVEfmt ", 1, 8, f) == 8 && fwrite(&fmt_len, 4, 1, f) == 1 &&
              fwrite(&pcm_tag, 2, 1, f) == 1 && fwrite(&ch, 2, 1, f) == 1 && fwrite(&sr, 4, 1, f) == 1 && fwrite(&byte_rate, 4, 1, f) == 1 &&
              fwrite(&align, 2, 1, f) == 1 && fwrite(&bits, 2, 1, f) == 1 && fwrite("data", 1, 4, f) == 4 && fwrite(&data_bytes, 4, 1, f) == 1 &&
              (n == 0 || fwrite(pcm.data(), 2, (size_t)n, f) == (size_t)n);
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : fail(std::string("short write to ") + path);
}

int fc_debug_freq_features(void* dev_buf, size_t cap_bytes, int mode) {
    if (mode < 0 || mode > 2 || (mode != 0 && !dev_buf)) return fail("fc_debug_freq_features: mode 0 (off), 1 (capture) or 2 (override) with a device buffer");
    g_feat_hook.buf = (float*)dev_buf; g_feat_hook.cap = cap_bytes; g_feat_hook.mode = mode;
    return 0;
}

// Host-side description of the conv kernel's operand layout for one chunk shape (kernels.hip: conv_pack_index / conv_koff_table); no GPU work.
int fc_debug_conv_layout(int k, int stride, int dil, int CC, int BM, int BN, int row, int* info /* [6] */, int* pack_index, size_t pack_cap,
                         int* koff, size_t koff_cap) {
    if (k < 1 || stride < 1 || dil < 1 || CC < 1 || BM < 1 || BN < 1 || !info) return fail("fc_debug_conv_layout: bad shape");
    const std::vector<int> t = fc::conv_koff_table(k, stride, dil, CC, BN, row ? 1 : 0);
    const int slabW = (BN - 1) * stride + (k - 1) * dil + 1;
    int PL = (slabW + stride - 1) / stride, rowStride = PL * stride;
    if (row) { rowStride = (slabW + 3) & ~3; PL = rowStride; }
    info[0] = fc::conv_quad(CC) ? 1 : 0; info[1] = fc::conv_wbuf_floats(k, CC, BM); info[2] = (int)t.size();
    info[3] = rowStride; info[4] = PL; info[5] = slabW;
    if (pack_index) {
        if (pack_cap < (size_t)k * CC * BM) return fail("fc_debug_conv_layout: pack_index buffer too small");
        for (int kk = 0; kk < k; ++kk)
            for (int cl = 0; cl < CC; ++cl)
                for (int mm = 0; mm < BM; ++mm) pack_index[((size_t)kk * CC + cl) * BM + mm] = (int)fc::conv_pack_index(k, CC, BM, kk, cl, mm);
    }
    if (koff) {
        if (koff_cap < t.size()) return fail("fc_debug_conv_layout: koff buffer too small");
        for (size_t i = 0; i < t.size(); ++i) koff[i] = t[i];
    }
    return 0;
}

int fc_debug_timeline(unsigned long long* dst) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(fc::debug_timeline(dst));
    return 0;
}

}  // extern "C"
