// LauraTTS generation engine: plan, checkpoint ingestion / weight re-layout, workspace planning and the fc_laura_* part of the
// C ABI (include/funcodec_amd.h).  Every Linear exists in up to two device layouts: the packed image of the codec's implicit-GEMM
// conv kernel (full-sequence form) and MFMA-fragment order (decoding step, LM only).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/funcodec_amd.h"
#include "kernels.h"
#include "laura_kernels.h"

// fc_last_error()'s thread-local message lives in engine.hip
extern "C" void fc_set_last_error_(const char* msg);

namespace {

namespace lk = fc::laura;

int fail(const std::string& msg) { fc_set_last_error_(msg.c_str()); return 1; }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

int ceil_div_i(int a, int b) { return (a + b - 1) / b; }
int pad4(int n) { return (n + 3) & ~3; }

struct HostTensor {
    std::vector<int64_t> dims;
    std::vector<float> data;
    bool set = false;
};

// One Linear of the checkpoint.
struct Lin {
    std::string name;       // state_dict prefix ("...linear_qkv" = rows of linear_q, linear_k, linear_v stacked)
    int cin = 0, cout = 0;
    // full-sequence form: k = 1 layer of conv_mfma_kernel
    int BM = 128, BN = 128, CC = 2, nchunk = 1, Mpad = 0;
    bool row = false;
    float *wt = nullptr, *bias = nullptr;
    int* koff = nullptr;
    // step form
    float* wf = nullptr;    // fragment order, null if the layer never runs in a decoding step
};

struct Block {
    Lin qkv, out, ff1, ff2;
    float *n1g = nullptr, *n1b = nullptr, *n2g = nullptr, *n2b = nullptr;   // attention norm, FFN norm
    float *bu = nullptr, *bv = nullptr;                                       // pos_bias_u / v flattened [d]
    float* ptab = nullptr;                                                    // [d][PR]
};

struct Stack {
    std::string prefix;
    fc_laura_stack s;
    Lin embed;
    float *eg = nullptr, *eb = nullptr, *ag = nullptr, *ab = nullptr;         // input-layer LayerNorm, after_norm
    std::vector<Block> blocks;
    bool step = false;
};

struct Ctx {
    int B = 0;
    hipStream_t st = nullptr;
    char* base = nullptr;
    size_t cap = 0, off = 0;
    bool dry = false;
    int err = 0;
    template <typename T>
    T* alloc(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = dry ? nullptr : (T*)(base + off);
        off += n * sizeof(T);
        if (!dry && off > cap) { err = 1; fail("workspace too small"); return nullptr; }
        return p;
    }
    bool live() const { return !dry && !err; }
    void check(hipError_t e, const char* what) {
        if (e != hipSuccess && !err) { err = 1; fail(std::string(what) + ": " + hipGetErrorString(e)); }
    }
};

}  // namespace

struct fc_laura {
    fc_laura_arch arch;
    int device = 0;
    bool finalized = false;
    int R = 0, PR = 0;      // relative positions -(R-1) .. R-1; row stride of the position tables
    std::vector<std::pair<std::string, std::vector<int64_t>>> expected;
    std::map<std::string, HostTensor> host;
    Stack text_encoder, codec_lm, codec_encoder;
    Lin text_out, lm_decoder, codec_out;
    float *lm_emb = nullptr, *cb = nullptr, *tok_emb = nullptr, *pe_abs = nullptr;
    float* lm_embed_wt = nullptr;      // codec_lm.encoder.embed.0.weight transposed [D][d] (the sampler's fused input layer)
    lk::StepLayer* step_layers = nullptr;   // device table of the LM's blocks for the persistent decoding step (laura_persist.hip)
    int persist_grid = 0;              // workgroups of the persistent step launch, 0 = not available on this device / for this model
    bool persist_on = !(getenv("FC_LAURA_PERSIST") && atoi(getenv("FC_LAURA_PERSIST")) == 0);   // fc_laura_set_persistent_step
    int persist_fallbacks = 0;         // calls that timed out at a hand-off of the persistent step and were re-run on the kernel chain
    std::map<std::string, Lin*> lin_by_name;
    std::vector<void*> dev_allocs;
    int vocab() const { return arch.predict_nq * (arch.codebook_size + 1); }
};

namespace {

template <typename T>
int upload(fc_laura* e, const std::vector<T>& h, T** out) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, h.size() * sizeof(T) + 16));
    HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    e->dev_allocs.push_back(d);
    *out = (T*)d;
    return 0;
}

// ---- plan ------------------------------------------------------------------------------------------------------------------
void expect(fc_laura* e, const std::string& name, std::vector<int64_t> dims) { e->expected.push_back({name, std::move(dims)}); }

void lin_tiling(Lin& L) {
    // the same choices engine.hip::choose_tiling makes for a k = 1, stride 1 layer whose input is already materialised
    if (L.cout > 64) { L.BM = 128; L.BN = 128; }
    else if (L.cout > 32) { L.BM = 64; L.BN = 256; }
    else { L.BM = 32; L.BN = 256; }
    const size_t lds_budget = (size_t)(160 / fc::conv_wgs_per_cu(L.BM)) * 1024;
    int cin_p2 = 2;
    while (cin_p2 < L.cin) cin_p2 *= 2;
    int cc = 2;
    for (;;) {
        const int n = cc * 2;
        if (n > 32 || n > cin_p2) break;
        if (!fc::conv_slab_fits(1, 1, 1, n, L.BN, L.BM, false)) break;
        if (fc::conv_lds_bytes_for(1, 1, 1, n, L.BM, L.BN, L.cin, 0, 0) > lds_budget) break;
        cc = n;
    }
    L.row = false;
    int best = 0;
    for (int n = 4; n <= 64 && n <= L.cin; n *= 2) {
        if (!fc::conv_row_ok(1, 1, 1, n, L.BM, L.BN, L.cin, false)) continue;
        if (fc::conv_lds_bytes_for(1, 1, 1, n, L.BM, L.BN, L.cin, 0, 1) > lds_budget) continue;
        best = n;
    }
    if (best > cc) { L.row = true; cc = best; }
    L.CC = cc;
    L.nchunk = ceil_div_i(L.cin, cc);
    L.Mpad = ceil_div_i(L.cout, L.BM) * L.BM;
}

Lin mk_lin(fc_laura* e, const std::string& name, int cin, int cout, bool bias, bool in_ckpt = true) {
    Lin L;
    L.name = name; L.cin = cin; L.cout = cout;
    lin_tiling(L);
    if (in_ckpt) {
        expect(e, name + ".weight", {cout, cin});
        if (bias) expect(e, name + ".bias", {cout});
    }
    return L;
}

void plan_stack(fc_laura* e, Stack& S, const std::string& prefix, const fc_laura_stack& s, bool step) {
    S.prefix = prefix; S.s = s; S.step = step;
    const int d = s.d_model, dk = d / s.heads;
    S.embed = mk_lin(e, prefix + ".embed.0", s.idim, d, true);
    expect(e, prefix + ".embed.1.weight", {d});
    expect(e, prefix + ".embed.1.bias", {d});
    const char* n_att = s.norm_style ? "norm1" : "norm_mha";
    const char* n_ff = s.norm_style ? "norm2" : "norm_ff";
    S.blocks.resize(s.layers);
    for (int i = 0; i < s.layers; ++i) {
        Block& b = S.blocks[i];
        const std::string p = prefix + ".encoders." + std::to_string(i);
        expect(e, p + ".self_attn.pos_bias_u", {s.heads, dk});
        expect(e, p + ".self_attn.pos_bias_v", {s.heads, dk});
        for (const char* n : {"linear_q", "linear_k", "linear_v"}) {
            expect(e, p + ".self_attn." + n + ".weight", {d, d});
            expect(e, p + ".self_attn." + n + ".bias", {d});
        }
        b.qkv = mk_lin(e, p + ".self_attn.linear_qkv", d, 3 * d, true, false);
        b.out = mk_lin(e, p + ".self_attn.linear_out", d, d, true);
        expect(e, p + ".self_attn.linear_pos.weight", {d, d});
        b.ff1 = mk_lin(e, p + ".feed_forward.w_1", d, s.ff, true);
        b.ff2 = mk_lin(e, p + ".feed_forward.w_2", s.ff, d, true);
        for (const char* n : {n_att, n_ff}) {
            expect(e, p + "." + n + ".weight", {d});
            expect(e, p + "." + n + ".bias", {d});
        }
    }
    expect(e, prefix + ".after_norm.weight", {d});
    expect(e, prefix + ".after_norm.bias", {d});
}

void register_lins(fc_laura* e) {
    auto reg = [&](Lin& L) { e->lin_by_name[L.name] = &L; };
    for (Stack* S : {&e->text_encoder, &e->codec_lm, &e->codec_encoder}) {
        reg(S->embed);
        for (Block& b : S->blocks) { reg(b.qkv); reg(b.out); reg(b.ff1); reg(b.ff2); }
    }
    reg(e->text_out); reg(e->lm_decoder); reg(e->codec_out);
}

// ---- packing ---------------------------------------------------------------------------------------------------------------
// W [cout][cin] row-major (torch Linear) -> conv_mfma_kernel's chunk images [m_tile][chunk][tap 0][local channel][BM rows]
int pack_full(fc_laura* e, Lin& L, const std::vector<float>& W, const std::vector<float>& bias) {
    const int mtiles = L.Mpad / L.BM;
    const int wbuf = fc::conv_wbuf_floats(1, L.CC, L.BM);
    std::vector<float> packed((size_t)mtiles * L.nchunk * wbuf, 0.f);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int ch = 0; ch < L.nchunk; ++ch)
            for (int cl = 0; cl < L.CC; ++cl) {
                const int ci = ch * L.CC + cl;
                if (ci >= L.cin) continue;
                float* img = &packed[(size_t)(mt * L.nchunk + ch) * wbuf];
                for (int mm = 0; mm < L.BM; ++mm) {
                    const int m = mt * L.BM + mm;
                    if (m < L.cout) img[fc::conv_pack_index(1, L.CC, L.BM, 0, cl, mm)] = W[(size_t)m * L.cin + ci];
                }
            }
    std::vector<float> bpad(L.Mpad, 0.f);
    for (int m = 0; m < L.cout && m < (int)bias.size(); ++m) bpad[m] = bias[m];
    if (upload(e, packed, &L.wt)) return 1;
    if (upload(e, bpad, &L.bias)) return 1;
    if (upload(e, fc::conv_koff_table(1, 1, 1, L.CC, L.BN, L.row ? 1 : 0), &L.koff)) return 1;
    return 0;
}

// fragment order of the decoding step's GEMV: [16-row tile][16-wide k chunk][lane = 16 g + r][j] = W[16 tile + r][16 chunk + 4 g + j]
int pack_step(fc_laura* e, Lin& L, const std::vector<float>& W) {
    if (L.cin % 16) return fail("decoding-step GEMV needs an input width that is a multiple of 16: " + L.name);
    const int tiles = ceil_div_i(L.cout, 16), nch = L.cin / 16;
    std::vector<float> f((size_t)tiles * nch * 256, 0.f);
    for (int t = 0; t < tiles; ++t)
        for (int c = 0; c < nch; ++c)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4, n = 16 * t + r;
                if (n >= L.cout) continue;
                for (int j = 0; j < 4; ++j)
                    f[((size_t)(t * nch + c) * 64 + lane) * 4 + j] = W[(size_t)n * L.cin + 16 * c + 4 * g + j];
            }
    return upload(e, f, &L.wf);
}

int pack_lin(fc_laura* e, Lin& L, bool step, const std::vector<float>* Wp = nullptr, const std::vector<float>* Bp = nullptr) {
    static const std::vector<float> none;
    const std::vector<float>& W = Wp ? *Wp : e->host[L.name + ".weight"].data;
    const std::vector<float>& Bv = Bp ? *Bp : (e->host.count(L.name + ".bias") ? e->host[L.name + ".bias"].data : none);
    if (pack_full(e, L, W, Bv)) return 1;
    if (step && pack_step(e, L, W)) return 1;
    return 0;
}

hipError_t launch_lin_full(const Lin& L, const float* in, int B, int T, float* out, hipStream_t st) {
    fc::ConvLaunch c;
    c.s0.ptr = in; c.s0.used = 1;
    c.wt = L.wt; c.bias = L.bias; c.koff = L.koff;
    c.out = out; c.out_sB = (long long)L.cout * T; c.out_sM = T; c.out_sT = 1;
    c.B = B; c.Cin = L.cin; c.Tin = T; c.M = L.cout; c.Tout = T;
    c.k = 1; c.stride = 1; c.padL = 0; c.padR = 0; c.dil = 1; c.pad_zero = 1;
    c.BM = L.BM; c.BN = L.BN; c.CC = L.CC; c.nchunk = L.nchunk; c.row = L.row ? 1 : 0;
    return fc::launch_conv(c, st);
}

// RelPositionalEncoding.extend_pe (funcodec/modules/embedding.py:293-306) evaluated the way torch evaluates it in fp32:
// div_term = exp(float(2m) * float(-(ln 10000 / d))), pe(r)[2m] = sin(r * div_term), pe(r)[2m+1] = cos(r * div_term); negative
// relative positions are sin / cos of the negated product.  Feature-major [d][PR], column = r + R - 1.
std::vector<float> rel_pe_fm(int d, int R, int PR) {
    std::vector<float> pe((size_t)d * PR, 0.f);
    for (int m = 0; m < d / 2; ++m) {
        const float div = expf((float)(2 * m) * (float)(-(std::log(10000.0) / (double)d)));
        for (int col = 0; col < PR; ++col) {
            const int r = col - (R - 1);
            const float a = (float)(r < 0 ? -r : r) * div;
            const float sn = sinf(a), cs = cosf(a);
            pe[(size_t)(2 * m) * PR + col] = r < 0 ? -sn : sn;
            pe[(size_t)(2 * m + 1) * PR + col] = cs;
        }
    }
    return pe;
}

int pack_stack(fc_laura* e, Stack& S, const float* pe_dev, void* ws, size_t ws_bytes) {
    (void)ws; (void)ws_bytes;
    const int d = S.s.d_model;
    auto H = [&](const std::string& k) -> const std::vector<float>& { return e->host[k].data; };
    if (pack_lin(e, S.embed, S.step)) return 1;
    if (upload(e, H(S.prefix + ".embed.1.weight"), &S.eg)) return 1;
    if (upload(e, H(S.prefix + ".embed.1.bias"), &S.eb)) return 1;
    if (upload(e, H(S.prefix + ".after_norm.weight"), &S.ag)) return 1;
    if (upload(e, H(S.prefix + ".after_norm.bias"), &S.ab)) return 1;
    const char* n_att = S.s.norm_style ? "norm1" : "norm_mha";
    const char* n_ff = S.s.norm_style ? "norm2" : "norm_ff";
    for (int i = 0; i < S.s.layers; ++i) {
        Block& b = S.blocks[i];
        const std::string p = S.prefix + ".encoders." + std::to_string(i);
        std::vector<float> W((size_t)3 * d * d), Bv((size_t)3 * d);
        int part = 0;
        for (const char* n : {"linear_q", "linear_k", "linear_v"}) {
            const auto& w = H(p + ".self_attn." + n + ".weight");
            const auto& bb = H(p + ".self_attn." + n + ".bias");
            std::copy(w.begin(), w.end(), W.begin() + (size_t)part * d * d);
            std::copy(bb.begin(), bb.end(), Bv.begin() + (size_t)part * d);
            ++part;
        }
        if (pack_lin(e, b.qkv, S.step, &W, &Bv)) return 1;
        if (pack_lin(e, b.out, S.step)) return 1;
        if (pack_lin(e, b.ff1, S.step)) return 1;
        if (pack_lin(e, b.ff2, S.step)) return 1;
        if (upload(e, H(p + "." + n_att + ".weight"), &b.n1g)) return 1;
        if (upload(e, H(p + "." + n_att + ".bias"), &b.n1b)) return 1;
        if (upload(e, H(p + "." + n_ff + ".weight"), &b.n2g)) return 1;
        if (upload(e, H(p + "." + n_ff + ".bias"), &b.n2b)) return 1;
        if (upload(e, H(p + ".self_attn.pos_bias_u"), &b.bu)) return 1;
        if (upload(e, H(p + ".self_attn.pos_bias_v"), &b.bv)) return 1;
        // position table of the block: linear_pos (no bias) applied to every relative-position encoding, on the device
        Lin pos;
        pos.name = p + ".self_attn.linear_pos"; pos.cin = d; pos.cout = d;
        lin_tiling(pos);
        if (pack_lin(e, pos, false)) return 1;
        void* tab = nullptr;
        HIP_TRY(hipMalloc(&tab, (size_t)d * e->PR * sizeof(float)));
        e->dev_allocs.push_back(tab);
        b.ptab = (float*)tab;
        HIP_TRY(launch_lin_full(pos, pe_dev, 1, e->PR, b.ptab, nullptr));
    }
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

// ---- execution ---------------------------------------------------------------------------------------------------------------
struct KvOut {            // LM prefix pass: K / V rows of every block go to the decoding step's caches
    float* kc = nullptr;  // [layers][B][d][Tcap]
    float* vc = nullptr;  // [layers][B][Tcap][d]
    int Tcap = 0;
};

// debugging aid (fc_laura_debug_probe): copy one intermediate tensor of the next full-sequence stack run to a caller buffer
struct Probe { float* dst = nullptr; size_t cap = 0; int layer = -1, what = -1, stack = -1; };
thread_local Probe g_probe;
void probe(Ctx& cx, const Stack& S, int what, int layer, const float* src, size_t n) {
    if (!g_probe.dst || !cx.live() || g_probe.what != what || g_probe.layer != layer) return;
    if (g_probe.stack >= 0 && g_probe.stack != (S.prefix == "text_encoder" ? 0 : S.prefix == "codec_lm.encoder" ? 1 : 2)) return;
    if (n * sizeof(float) > g_probe.cap) n = g_probe.cap / sizeof(float);
    cx.check(hipMemcpyAsync(g_probe.dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, cx.st), "probe");
}

// The stack on feature-major input [B][idim][T]; returns after_norm(x) [B][d][T].
float* run_stack_full(fc_laura* e, Ctx& cx, const Stack& S, const float* in, int T, const int* lens, const int* bidir, int causal,
                      const KvOut* kv) {
    const int B = cx.B, d = S.s.d_model, ff = S.s.ff;
    float* x = cx.alloc<float>((size_t)B * d * T);
    float* xn = cx.alloc<float>((size_t)B * d * T);
    float* dl = cx.alloc<float>((size_t)B * d * T);
    float* qkv = cx.alloc<float>((size_t)B * 3 * d * T);
    float* ctx = cx.alloc<float>((size_t)B * d * T);
    float* hb = cx.alloc<float>((size_t)B * ff * T);
    if (!cx.live()) return xn;
    cx.check(launch_lin_full(S.embed, in, B, T, dl, cx.st), "embed GEMM");
    // Linear -> LayerNorm(eps 1e-5) -> [ReLU] -> x * sqrt(d)   (conformer_encoder.py:372-378, transformer_encoder.py:463-470)
    cx.check(lk::launch_layernorm_fm(dl, nullptr, nullptr, S.eg, S.eb, 1e-5f, S.s.embed_relu, sqrtf((float)d), x, B, d, T, cx.st), "embed LayerNorm");
    probe(cx, S, 0, 0, x, (size_t)B * d * T);
    const float* pending = nullptr;
    for (int i = 0; i < S.s.layers && !cx.err; ++i) {
        const Block& b = S.blocks[i];
        cx.check(lk::launch_layernorm_fm(x, pending, pending ? x : nullptr, b.n1g, b.n1b, 1e-12f, 0, 1.f, xn, B, d, T, cx.st), "LayerNorm");
        probe(cx, S, 5, i, x, (size_t)B * d * T);
        probe(cx, S, 1, i, xn, (size_t)B * d * T);
        cx.check(launch_lin_full(b.qkv, xn, B, T, qkv, cx.st), "QKV GEMM");
        probe(cx, S, 2, i, qkv, (size_t)B * 3 * d * T);
        if (kv) cx.check(lk::launch_kv_store(qkv, lens, B, d, T, kv->Tcap, kv->kc + (size_t)i * B * d * kv->Tcap,
                                             kv->vc + (size_t)i * B * d * kv->Tcap, cx.st), "kv store");
        lk::AttnFull a;
        a.qkv = qkv; a.ptab = b.ptab; a.bias_u = b.bu; a.bias_v = b.bv; a.lens = lens; a.bidir = bidir; a.causal = causal; a.ctx = ctx;
        a.B = B; a.H = S.s.heads; a.DK = d / S.s.heads; a.T = T; a.R = e->R; a.PR = e->PR;
        cx.check(lk::launch_attn_full(a, cx.st), "attention");
        probe(cx, S, 3, i, ctx, (size_t)B * d * T);
        cx.check(launch_lin_full(b.out, ctx, B, T, dl, cx.st), "out GEMM");
        cx.check(lk::launch_layernorm_fm(x, dl, x, b.n2g, b.n2b, 1e-12f, 0, 1.f, xn, B, d, T, cx.st), "LayerNorm");
        cx.check(launch_lin_full(b.ff1, xn, B, T, hb, cx.st), "FFN GEMM 1");
        cx.check(lk::launch_act(hb, (size_t)B * ff * T, S.s.act, cx.st), "activation");
        cx.check(launch_lin_full(b.ff2, hb, B, T, dl, cx.st), "FFN GEMM 2");
        pending = dl;
    }
    cx.check(lk::launch_layernorm_fm(x, pending, pending ? x : nullptr, S.ag, S.ab, 1e-12f, 0, 1.f, xn, B, d, T, cx.st), "after_norm");
    probe(cx, S, 5, S.s.layers, x, (size_t)B * d * T);
    return xn;
}

struct LenPack { int v[64]; };
__global__ void set_lens_kernel(LenPack p, int* dst, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = p.v[threadIdx.x];
}

// Lengths travel BY VALUE in kernel arguments (64 per launch): the caller's arrays are temporaries of the binding, and an asynchronous
// copy from pageable host memory is allowed to read them after the call has returned.
int* upload_lens(Ctx& cx, const int32_t* host, int B, int fill = 0) {
    int* d = cx.alloc<int>(B);
    if (!cx.live()) return d;
    if (host) {
        for (int b0 = 0; b0 < B; b0 += 64) {
            LenPack p;
            const int n = B - b0 < 64 ? B - b0 : 64;
            for (int i = 0; i < 64; ++i) p.v[i] = i < n ? host[b0 + i] : 0;
            hipLaunchKernelGGL(set_lens_kernel, dim3(1), dim3(64), 0, cx.st, p, d + b0, n);
        }
        cx.check(hipGetLastError(), "lengths upload");
    } else cx.check(lk::launch_fill_i32(d, fill, B, cx.st), "fill");
    return d;
}

int check_lens(const int32_t* lens, int B, int lo, int hi, const char* what) {
    for (int b = 0; b < B; ++b)
        if (lens[b] < lo || lens[b] > hi) return fail(std::string(what) + " out of range");
    return 0;
}

Ctx make_ctx(int B, void* ws, size_t ws_bytes, void* stream) {
    Ctx cx;
    cx.B = B; cx.st = (hipStream_t)stream; cx.base = (char*)ws; cx.cap = ws_bytes;
    cx.dry = ws == nullptr;
    return cx;
}

int max_of(const int32_t* v, int n) {
    int m = 0;
    for (int i = 0; i < n; ++i) m = std::max(m, (int)v[i]);
    return m;
}

// text side: token-major text_outs [B][L][D] -> feature-major [B][D][Tt]
float* text_to_fm(fc_laura* e, Ctx& cx, const float* text_outs, const int* tl_dev, int L, int* Tt_out) {
    const int D = e->arch.codebook_dim, Tt = pad4(L);
    float* fm = cx.alloc<float>((size_t)cx.B * D * Tt);
    if (cx.live()) cx.check(lk::launch_tm_to_fm(text_outs, tl_dev, cx.B, L, D, Tt, fm, cx.st), "text transpose");
    *Tt_out = Tt;
    return fm;
}

int do_encode(fc_laura* e, Ctx& cx, const float* text_emb, const int64_t* text_ids, const int32_t* text_lens, int L, float* text_outs) {
    const int B = cx.B, idim = e->arch.input_size, D = e->arch.codebook_dim, T = pad4(L);
    int* tl = upload_lens(cx, text_lens, B);
    float* in = cx.alloc<float>((size_t)B * idim * T);
    if (cx.live()) {
        if (text_ids) cx.check(lk::launch_token_embed_fm(text_ids, e->tok_emb, e->arch.vocab_size, B, L, idim, T, in, cx.st), "token embedding");
        else cx.check(lk::launch_tm_to_fm(text_emb, tl, B, L, idim, T, in, cx.st), "text transpose");
    }
    float* h = run_stack_full(e, cx, e->text_encoder, in, T, tl, nullptr, 0, nullptr);
    float* o = cx.alloc<float>((size_t)B * D * T);
    if (cx.live()) {
        cx.check(launch_lin_full(e->text_out, h, B, T, o, cx.st), "text_enc_out_layer");
        cx.check(lk::launch_fm_to_tm(o, tl, B, D, T, L, text_outs, cx.st), "output transpose");
    }
    return cx.err;
}

int do_lm_logprobs(fc_laura* e, Ctx& cx, const float* text_outs, const int32_t* text_lens, int L, const int64_t* codec,
                   const int32_t* codec_lens, int Cmax, float* logp, int Tseq) {
    const int B = cx.B, D = e->arch.codebook_dim, V = e->vocab();
    const int T = pad4(Tseq);
    int* tl = upload_lens(cx, text_lens, B);
    int* cl = upload_lens(cx, codec ? codec_lens : nullptr, B, 0);
    int Tt = 0;
    float* tfm = text_to_fm(e, cx, text_outs, tl, L, &Tt);
    float* seq = cx.alloc<float>((size_t)B * D * T);
    int* seq_len = cx.alloc<int>(B);
    int* bidir = cx.alloc<int>(B);
    if (cx.live())
        cx.check(lk::launch_lm_assemble(tfm, Tt, tl, e->lm_emb, e->cb, e->arch.codebook_size, e->arch.predict_nq, codec, cl, Cmax, B, D, T,
                                        seq, seq_len, bidir, cx.st), "LM input");
    float* h = run_stack_full(e, cx, e->codec_lm, seq, T, seq_len, e->arch.bidirectional_inputs ? bidir : nullptr, 1, nullptr);
    float* lg = cx.alloc<float>((size_t)B * V * T);
    if (cx.live()) {
        cx.check(launch_lin_full(e->lm_decoder, h, B, T, lg, cx.st), "LM decoder GEMM");
        cx.check(lk::launch_logsoftmax_fm(lg, seq_len, B, V, T, Tseq, logp, cx.st), "log-softmax");
    }
    return cx.err;
}

int do_codec_emb(fc_laura* e, Ctx& cx, const float* text_outs, const int32_t* text_lens, int L, const int64_t* codec, int nq_cols,
                 const int32_t* codec_lens, int Cmax, float* emb, int Tmax) {
    const int B = cx.B, D = e->arch.codebook_dim, T = pad4(Tmax);
    int* tl = upload_lens(cx, text_lens, B);
    int* cl = upload_lens(cx, codec_lens, B);
    int Tt = 0;
    float* tfm = text_to_fm(e, cx, text_outs, tl, L, &Tt);
    float* in = cx.alloc<float>((size_t)B * D * T);
    int* seq_len = cx.alloc<int>(B);
    if (cx.live())
        cx.check(lk::launch_nar_assemble(tfm, Tt, tl, e->cb, e->arch.codebook_size, e->arch.predict_nq, codec, cl, Cmax, nq_cols, e->pe_abs,
                                         e->arch.pos_emb_split, B, D, T, in, seq_len, cx.st), "predictor input");
    float* h = run_stack_full(e, cx, e->codec_encoder, in, T, seq_len, nullptr, 0, nullptr);
    float* o = cx.alloc<float>((size_t)B * D * T);
    if (cx.live()) {
        cx.check(launch_lin_full(e->codec_out, h, B, T, o, cx.st), "codec_encoder_out_layer");
        cx.check(lk::launch_nar_extract(o, tl, cl, B, D, T, Cmax, emb, cx.st), "predictor output");
    }
    return cx.err;
}

hipError_t step_gemv(const Lin& L, const float* x, int B, const float* gamma, const float* beta, float eps, int act, int mode, float* y,
                     int ldy, hipStream_t st, float* kc = nullptr, float* vc = nullptr, const int* pos = nullptr, int d = 0, int Tcap = 0,
                     const float* apart = nullptr, int H = 0, int DK = 0, int NS = 0) {
    lk::Gemv g;
    g.apart = apart; g.H = H; g.DK = DK; g.NS = NS;
    g.x = x; g.wf = L.wf; g.bias = L.bias; g.gamma = gamma; g.beta = beta; g.eps = eps; g.act = act; g.mode = mode; g.y = y; g.ldy = ldy;
    g.kc = kc; g.vc = vc; g.pos = pos; g.d = d; g.Tcap = Tcap; g.B = B; g.K = L.cin; g.N = L.cout;
    return lk::launch_gemv(g, st);
}

__global__ void copy_prompt_kernel(const int64_t* src, const int* lens, int Cmax, int nq, int64_t* dst, int stride) {
    const int b = blockIdx.x;
    const int n = lens[b] * nq;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[(size_t)b * stride * nq + i] = src[(size_t)b * Cmax * nq + i];
}

int do_decode(fc_laura* e, Ctx& cx, const float* text_outs, const int32_t* text_lens, int L, const int64_t* continual,
              const int32_t* cont_lens, int Cmax, int max_length, int mode, int ki, float pf, uint64_t seed, const int64_t* forced,
              int64_t* tokens, int32_t* out_lens, float* step_logp, int max_prefix) {
    const Stack& S = e->codec_lm;
    const int B = cx.B, D = e->arch.codebook_dim, V = e->vocab(), d = S.s.d_model, ff = S.s.ff, NL = S.s.layers;
    const int nq = e->arch.predict_nq, K = e->arch.codebook_size;
    const int T = pad4(max_prefix);
    const int Tcap = pad4(max_prefix + max_length);
    int* tl = upload_lens(cx, text_lens, B);
    int* cl = upload_lens(cx, continual ? cont_lens : nullptr, B, 0);
    int Tt = 0;
    float* tfm = text_to_fm(e, cx, text_outs, tl, L, &Tt);
    float* seq = cx.alloc<float>((size_t)B * D * T);
    int* pos = cx.alloc<int>(B);            // = sequence length after the prefix pass: the cache fill
    int* bidir = cx.alloc<int>(B);
    int* ctr = cx.alloc<int>(3 * B + 4);    // n_gen[B], done[B], step[B], n_done
    KvOut kv;
    kv.Tcap = Tcap;
    kv.kc = cx.alloc<float>((size_t)NL * B * d * Tcap);
    kv.vc = cx.alloc<float>((size_t)NL * B * d * Tcap);
    float* xs = cx.alloc<float>((size_t)16 * d);
    float* qb = cx.alloc<float>((size_t)16 * d);
    const int heads = S.s.heads, dkh = d / S.s.heads;
    int NS = 256 / (B * heads);                 // key ranges per (utterance, head): fill the chip's 256 CUs
    NS = NS < 1 ? 1 : (NS > 8 ? 8 : NS);
    float* apart = cx.alloc<float>((size_t)16 * heads * 8 * (dkh + 2));
    float* hb = cx.alloc<float>((size_t)16 * ff);
    float* lg = cx.alloc<float>((size_t)16 * V);
    float* nemb = cx.alloc<float>((size_t)16 * D);
    // the decoding step as ONE persistent launch (laura_persist.hip): per-block edge buffers, arrival counters, the launch counter
    const int KS2 = lk::step_persist_ksplit(d, ff);
    auto pad32 = [](size_t n) { return (n + 31) & ~(size_t)31; };
    const size_t o_q = 0, o_ap = o_q + pad32((size_t)16 * d), o_xm = o_ap + pad32((size_t)16 * heads * 8 * (dkh + 4)),
                 o_hb = o_xm + pad32((size_t)16 * d), o_xo = o_hb + pad32((size_t)16 * ff), edge_stride = o_xo + pad32((size_t)KS2 * 16 * d);
    float* edge = cx.alloc<float>(edge_stride * NL);
    const size_t sync_words = lk::step_persist_sync_words(NL);
    unsigned* psync = cx.alloc<unsigned>(sync_words + 32);
    unsigned* pseq = psync ? psync + sync_words : nullptr;          // one word behind the counters
    static const char* trace_path = getenv("FC_LAURA_TRACE");        // tuning aid: s_memtime stamps of the LAST persistent step -> file
    unsigned long long* ptrace = trace_path ? cx.alloc<unsigned long long>((size_t)256 * 64 * 8) : nullptr;
    const bool persist = e->persist_on && e->persist_grid > 0 && lk::step_persist_supported(B, d, ff, heads, dkh, V, NS);
    if (cx.live()) {
        cx.check(lk::launch_lm_assemble(tfm, Tt, tl, e->lm_emb, e->cb, K, nq, continual, cl, Cmax, B, D, T, seq, pos, bidir, cx.st), "LM input");
        cx.check(lk::launch_fill_i32(ctr, 0, 3 * B + 4, cx.st), "counters");
        cx.check(lk::launch_fill_i32((int*)psync, 0, (int)sync_words + 32, cx.st), "arrival counters");
        if (continual) hipLaunchKernelGGL(copy_prompt_kernel, dim3(B), dim3(256), 0, cx.st, continual, cl, Cmax, nq, tokens, Cmax + max_length);
    }
    float* h = run_stack_full(e, cx, S, seq, T, pos, e->arch.bidirectional_inputs ? bidir : nullptr, 1, &kv);
    if (cx.dry || cx.err) return cx.err;
    int *n_gen = ctr, *done = ctr + B, *step = ctr + 2 * B, *n_done = ctr + 3 * B;
    cx.check(lk::launch_gather_last(h, pos, B, d, T, xs, cx.st), "last position");
    cx.check(step_gemv(e->lm_decoder, xs, B, nullptr, nullptr, 0.f, 0, 0, lg, V, cx.st), "decoder GEMV");
    lk::Sample sm;
    sm.logits = lg; sm.B = B; sm.K = K; sm.nq = nq; sm.mode = mode; sm.ki = ki; sm.pf = pf; sm.seed = seed; sm.forced = forced;
    sm.max_steps = max_length; sm.tokens = tokens; sm.tok_stride = Cmax + max_length; sm.tok_off = cl; sm.n_gen = n_gen; sm.done = done;
    sm.n_done = n_done; sm.pos = pos; sm.step = step; sm.logp_out = step_logp; sm.cb = e->cb; sm.D = D; sm.next_emb = nemb;
    // the sampler also runs the LM's input layer on the new token (Linear + LayerNorm + ReLU + x * sqrt(d)): xs is ready for block 0
    sm.emb_wt = e->lm_embed_wt; sm.emb_bias = S.embed.bias; sm.emb_g = S.eg; sm.emb_b = S.eb; sm.dm = d; sm.emb_relu = S.s.embed_relu;
    sm.xscale = sqrtf((float)d); sm.xs = xs;
    sm.launch_seq = pseq;                  // every sampler launch numbers the persistent step launch that follows it (1, 2, ...)
    cx.check(lk::launch_sample(sm, cx.st), "sampling");
    int host_done = 0;
    lk::StepPersistArgs pa{};
    pa.layers = e->step_layers; pa.wdec = e->lm_decoder.wf; pa.bdec = e->lm_decoder.bias; pa.ag = S.ag; pa.ab = S.ab;
    pa.xs = xs; pa.logits = lg; pa.edge = edge; pa.kc = kv.kc; pa.vc = kv.vc; pa.pos = pos; pa.sync = psync; pa.seq = pseq;
    pa.edge_stride = edge_stride; pa.o_q = (int)o_q; pa.o_ap = (int)o_ap; pa.o_xm = (int)o_xm; pa.o_hb = (int)o_hb; pa.o_xo = (int)o_xo;
    pa.B = B; pa.d = d; pa.ff = ff; pa.H = heads; pa.DK = dkh; pa.NL = NL; pa.V = V; pa.Tcap = Tcap; pa.R = e->R; pa.PR = e->PR; pa.NS = NS;
    pa.act = S.s.act; pa.G = e->persist_grid; pa.KS2 = KS2; pa.trace = ptrace;
    { const char* t = getenv("FC_LAURA_PERSIST_TEST"); pa.test_timeout = t && std::string(t) == "timeout"; }
    if (ptrace && cx.live()) cx.check(hipMemsetAsync(ptrace, 0, (size_t)256 * 64 * 8 * sizeof(unsigned long long), cx.st), "trace");
    // one decoding step: the newest token of every utterance through the LM against its KV cache.  Every kernel reads its
    // positions from device memory, so the launch sequence is identical from step to step.
    auto run_step = [&](hipStream_t st) {
        if (persist) {
            cx.check(lk::launch_step_persist(pa, st), "persistent decoding step");
            cx.check(lk::launch_sample(sm, st), "sampling");
            return;
        }
        for (int i = 0; i < NL; ++i) {
            const Block& b = S.blocks[i];
            float* kc = kv.kc + (size_t)i * B * d * Tcap;
            float* vc = kv.vc + (size_t)i * B * d * Tcap;
            cx.check(step_gemv(b.qkv, xs, B, b.n1g, b.n1b, 1e-12f, 0, 2, qb, d, st, kc, vc, pos, d, Tcap), "QKV GEMV");
            lk::AttnStep a;
            a.q = qb; a.kc = kc; a.vc = vc; a.ptab = b.ptab; a.bias_u = b.bu; a.bias_v = b.bv; a.pos = pos; a.part = apart; a.NS = NS;
            a.B = B; a.H = heads; a.DK = dkh; a.Tcap = Tcap; a.R = e->R; a.PR = e->PR;
            cx.check(lk::launch_attn_step(a, st), "step attention");
            // linear_out reads the key-range partials and combines them while staging its input
            cx.check(step_gemv(b.out, nullptr, B, nullptr, nullptr, 0.f, 0, 1, xs, d, st, nullptr, nullptr, nullptr, 0, 0, apart, heads, dkh, NS), "out GEMV");
            cx.check(step_gemv(b.ff1, xs, B, b.n2g, b.n2b, 1e-12f, S.s.act, 0, hb, ff, st), "FFN GEMV 1");
            cx.check(step_gemv(b.ff2, hb, B, nullptr, nullptr, 0.f, 0, 1, xs, d, st), "FFN GEMV 2");
        }
        cx.check(step_gemv(e->lm_decoder, xs, B, S.ag, S.ab, 1e-12f, 0, 0, lg, V, st), "decoder GEMV");
        cx.check(lk::launch_sample(sm, st), "sampling");
    };
    bool timed_out = false;
    auto all_done = [&]() -> bool {       // all utterances finished? (one small read-back)
        unsigned perr = 0;             // both words in the same stream-ordered read-back: one synchronisation, no null-stream copy
        hipError_t ea = hipMemcpyAsync(&host_done, n_done, sizeof(int), hipMemcpyDeviceToHost, cx.st);
        if (ea == hipSuccess && persist) ea = hipMemcpyAsync(&perr, psync + sync_words - 64, sizeof(unsigned), hipMemcpyDeviceToHost, cx.st);
        if (ea != hipSuccess || hipStreamSynchronize(cx.st) != hipSuccess) { cx.err = 1; fail("decode_codec: status read-back failed"); return true; }
        if (perr) {                    // stop replaying steps once a hand-off has timed out; the caller re-runs the call on the kernel chain
            cx.err = 1;
            timed_out = true;
            return true;
        }
        return host_done >= B;
    };
    int s = 1;
    if (s < max_length && !cx.err) { run_step(cx.st); ++s; }          // first step eagerly (also sets the kernels' LDS attributes)
    // the remaining steps replay ONE captured HIP graph of the step (62 small launches): the loop is launch-bound otherwise
    static const bool graph_env = !(getenv("FC_LAURA_GRAPH") && atoi(getenv("FC_LAURA_GRAPH")) == 0);
    hipGraphExec_t gexec = nullptr;
    if (graph_env && cx.st != nullptr && max_length - s >= 4 && !cx.err) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(cx.st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            run_step(cx.st);
            const hipError_t ec = hipStreamEndCapture(cx.st, &graph);
            if (ec != hipSuccess || cx.err || hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) != hipSuccess) gexec = nullptr;
            if (graph) (void)hipGraphDestroy(graph);
            if (cx.err) return 1;                                       // a launch failed while capturing
            (void)hipGetLastError();
        }
    }
    for (; s < max_length && !cx.err; ++s) {
        if (gexec) cx.check(hipGraphLaunch(gexec, cx.st), "graph launch");
        else run_step(cx.st);
        if ((s & 15) == 15 && s + 1 < max_length && all_done()) break;
    }
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (timed_out) {
        (void)hipStreamSynchronize(cx.st);
        e->persist_on = false;
        return 2;
    }
    if (cx.err) return 1;
    if (persist && ptrace) {
        std::vector<unsigned long long> tr((size_t)256 * 64 * 8);
        HIP_TRY(hipMemcpyAsync(tr.data(), ptrace, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, cx.st));
        HIP_TRY(hipStreamSynchronize(cx.st));
        if (FILE* f = fopen(trace_path, "wb")) { fwrite(tr.data(), sizeof(unsigned long long), tr.size(), f); fclose(f); }
    }
    if (persist) {      // a hand-off of the persistent step timed out (a workgroup not resident, a lost store): never a silent result
        unsigned perr = 0;
        HIP_TRY(hipMemcpyAsync(&perr, psync + sync_words - 64, sizeof(unsigned), hipMemcpyDeviceToHost, cx.st));
        HIP_TRY(hipStreamSynchronize(cx.st));
        if (perr) {
            e->persist_on = false;         // never a silent result: the caller (fc_laura_decode_codec) re-runs this call on the kernel chain
            return 2;
        }
    }
    std::vector<int> gen(B);
    HIP_TRY(hipMemcpyAsync(gen.data(), n_gen, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, cx.st));
    HIP_TRY(hipStreamSynchronize(cx.st));
    for (int b = 0; b < B; ++b) out_lens[b] = (continual ? cont_lens[b] : 0) + gen[b];
    return 0;
}

int check_ready(fc_laura* e) {
    if (!e) return fail("null engine");
    if (!e->finalized) return fail("engine not finalized");
    HIP_TRY(hipSetDevice(e->device));
    return 0;
}

int check_stack(const fc_laura_stack& s, const char* name, bool step = false) {
    if (s.idim < 1 || s.d_model < 64 || s.heads < 1 || s.ff < 16 || s.layers < 1) return fail(std::string(name) + ": bad sizes");
    if (s.d_model % s.heads) return fail(std::string(name) + ": d_model must be a multiple of heads");
    const int dk = s.d_model / s.heads;
    if (dk != 32 && dk != 64) return fail(std::string(name) + ": head dimension must be 32 or 64");
    if (s.d_model % 16 || s.ff % 16) return fail(std::string(name) + ": d_model and ff must be multiples of 16");
    if (s.act != 1 && s.act != 2) return fail(std::string(name) + ": activation must be 1 (relu) or 2 (swish)");
    // limits of the kernels, refused HERE with a reason instead of a hipErrorInvalidValue at the first call: the LayerNorm kernels
    // and the sampler's fused input layer hold a row of <= 1024 channels; the step form's GEMV stages (16 + 1) x (K + 4) floats of input
    // plus 16 KiB of partial tiles in the 160 KiB of LDS, K = ff for w_2
    if (s.d_model > 1024) return fail(std::string(name) + ": d_model above 1024 is not supported (LayerNorm / sampler kernels)");
    if (step && ((size_t)17 * ((size_t)s.ff + 4) * 4 + 16 * 1024 > 160 * 1024))
        return fail(std::string(name) + ": feed-forward width too large for the decoding step's GEMV (LDS): ff <= 2112");
    return 0;
}

}  // namespace

extern "C" {

int fc_laura_create(const fc_laura_arch* arch, int device, fc_laura** out) {
    if (!arch || !out) return fail("null argument");
    if (arch->abi_version != FC_ABI_VERSION) return fail("fc_laura_arch.abi_version mismatch");
    if (arch->codebook_size != 1024) return fail("codebook_size must be 1024 (the reference's codec_index_shift is hard-wired to it)");
    if (arch->predict_nq < 1 || arch->predict_nq > 8 || arch->predict_nq > arch->num_quantizers) return fail("bad predict_nq");
    if (arch->codebook_dim % 16 || arch->codebook_dim < 16) return fail("codebook_dim must be a multiple of 16");
    if (arch->max_positions < 16 || arch->max_positions > 2048) return fail("max_positions must be in [16, 2048]");
    if (arch->max_positions % 4) return fail("max_positions must be a multiple of 4 (time axes are padded to whole quads)");
    if (check_stack(arch->text_encoder, "text_encoder") || check_stack(arch->codec_lm, "codec_lm", true) ||
        check_stack(arch->codec_encoder, "codec_encoder")) return 1;
    if (arch->codec_lm.idim != arch->codebook_dim || arch->codec_encoder.idim != arch->codebook_dim)
        return fail("codec_lm / codec_encoder input width must equal codebook_dim");
    if (arch->text_encoder.idim != arch->input_size) return fail("text_encoder input width must equal input_size");
    auto e = std::make_unique<fc_laura>();
    e->arch = *arch;
    e->device = device;
    e->R = arch->max_positions;
    e->PR = pad4(2 * e->R - 1);
    const int D = arch->codebook_dim;
    plan_stack(e.get(), e->text_encoder, "text_encoder", arch->text_encoder, false);
    e->text_out = mk_lin(e.get(), "text_enc_out_layer", arch->text_encoder.d_model, D, true);
    if (arch->vocab_size > 0) expect(e.get(), "token_embedding.weight", {arch->vocab_size, arch->input_size});
    expect(e.get(), "lm_embedding.weight", {2, D});
    plan_stack(e.get(), e->codec_lm, "codec_lm.encoder", arch->codec_lm, true);
    e->lm_decoder = mk_lin(e.get(), "codec_lm.decoder", arch->codec_lm.d_model, e->vocab(), true);
    plan_stack(e.get(), e->codec_encoder, "codec_encoder", arch->codec_encoder, false);
    e->codec_out = mk_lin(e.get(), "codec_encoder_out_layer", arch->codec_encoder.d_model, D, true);
    expect(e.get(), "quantizer_codebook.embed", {arch->num_quantizers, arch->codebook_size, D});
    register_lins(e.get());
    *out = e.release();
    return 0;
}

void fc_laura_destroy(fc_laura* e) {
    if (!e) return;
    for (void* p : e->dev_allocs) (void)hipFree(p);
    delete e;
}

int fc_laura_num_weights(const fc_laura* e) { return e ? (int)e->expected.size() : 0; }

int fc_laura_weight_info(const fc_laura* e, int i, const char** name, int64_t* dims) {
    if (!e || i < 0 || i >= (int)e->expected.size()) return -1;
    *name = e->expected[i].first.c_str();
    const auto& d = e->expected[i].second;
    for (size_t k = 0; k < d.size(); ++k) dims[k] = d[k];
    return (int)d.size();
}

int fc_laura_set_weight(fc_laura* e, const char* name, const float* host, const int64_t* dims, int ndim) {
    if (!e || !name || !host) return fail("null argument");
    for (const auto& ex : e->expected) {
        if (ex.first != name) continue;
        size_t n = 1;
        bool ok = (int)ex.second.size() == ndim;
        for (int k = 0; ok && k < ndim; ++k) { ok = ex.second[k] == dims[k]; n *= (size_t)dims[k]; }
        if (!ok) return fail(std::string("shape mismatch for ") + name);
        HostTensor& t = e->host[name];
        t.dims.assign(dims, dims + ndim);
        t.data.assign(host, host + n);
        t.set = true;
        return 0;
    }
    fail(std::string("unknown weight: ") + name);
    return 2;
}

int fc_laura_finalize(fc_laura* e) {
    if (!e) return fail("null engine");
    if (e->finalized) return 0;
    for (const auto& ex : e->expected)
        if (!e->host.count(ex.first) || !e->host[ex.first].set) return fail("missing weight: " + ex.first);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= e->device) return fail("no HIP device: the engine has no CPU path");
    HIP_TRY(hipSetDevice(e->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, e->device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return fail(std::string("gfx950 (MI355X) required, found ") + prop.gcnArchName);
    // relative-position encodings, one table per distinct model width
    std::map<int, float*> pe_dev;
    for (Stack* S : {&e->text_encoder, &e->codec_lm, &e->codec_encoder}) {
        const int d = S->s.d_model;
        if (!pe_dev.count(d)) {
            float* p = nullptr;
            if (upload(e, rel_pe_fm(d, e->R, e->PR), &p)) return 1;
            pe_dev[d] = p;
        }
        if (pack_stack(e, *S, pe_dev[d], nullptr, 0)) return 1;
    }
    {
        const auto& W = e->host["codec_lm.encoder.embed.0.weight"].data;       // [d][D]
        const int d = e->arch.codec_lm.d_model, D = e->arch.codebook_dim;
        std::vector<float> wt((size_t)D * d);
        for (int n = 0; n < d; ++n)
            for (int k = 0; k < D; ++k) wt[(size_t)k * d + n] = W[(size_t)n * D + k];
        if (upload(e, wt, &e->lm_embed_wt)) return 1;
    }
    if (pack_lin(e, e->text_out, false)) return 1;
    if (pack_lin(e, e->lm_decoder, true)) return 1;
    if (pack_lin(e, e->codec_out, false)) return 1;
    if (upload(e, e->host["lm_embedding.weight"].data, &e->lm_emb)) return 1;
    if (upload(e, e->host["quantizer_codebook.embed"].data, &e->cb)) return 1;
    if (e->arch.vocab_size > 0 && upload(e, e->host["token_embedding.weight"].data, &e->tok_emb)) return 1;
    {   // PositionalEncoding table of width codebook_dim (embedding.py:63-77), token-major [R][D]
        const int D = e->arch.codebook_dim;
        std::vector<float> pe((size_t)e->R * D);
        for (int m = 0; m < D / 2; ++m) {
            const float div = expf((float)(2 * m) * (float)(-(std::log(10000.0) / (double)D)));
            for (int r = 0; r < e->R; ++r) {
                pe[(size_t)r * D + 2 * m] = sinf((float)r * div);
                pe[(size_t)r * D + 2 * m + 1] = cosf((float)r * div);
            }
        }
        if (upload(e, pe, &e->pe_abs)) return 1;
    }
    {   // the persistent decoding step's view of the LM blocks
        const Stack& S = e->codec_lm;
        std::vector<lk::StepLayer> tab(S.blocks.size());
        for (size_t i = 0; i < S.blocks.size(); ++i) {
            const Block& b = S.blocks[i];
            tab[i] = lk::StepLayer{b.qkv.wf, b.qkv.bias, b.out.wf, b.out.bias, b.ff1.wf, b.ff1.bias, b.ff2.wf, b.ff2.bias,
                                   b.n1g, b.n1b, b.n2g, b.n2b, b.bu, b.bv, b.ptab};
        }
        if (upload(e, tab, &e->step_layers)) return 1;
        e->persist_grid = lk::step_persist_grid(e->device, S.s.d_model, S.s.ff);
    }
    e->host.clear();
    e->finalized = true;
    return 0;
}

size_t fc_laura_workspace_bytes(const fc_laura* ce, int B, int L, int Cmax, int max_length) {
    fc_laura* e = const_cast<fc_laura*>(ce);
    if (!e || B < 1 || L < 1) return 0;
    size_t need = 0;
    std::vector<int32_t> tl(B, L), cl(B, Cmax);
    {
        Ctx cx = make_ctx(B, nullptr, 0, nullptr);
        do_encode(e, cx, nullptr, nullptr, tl.data(), L, nullptr);
        need = std::max(need, cx.off);
    }
    {
        Ctx cx = make_ctx(B, nullptr, 0, nullptr);
        do_lm_logprobs(e, cx, nullptr, tl.data(), L, nullptr, cl.data(), Cmax, nullptr, L + 2 + Cmax + max_length);
        need = std::max(need, cx.off);
    }
    {
        Ctx cx = make_ctx(B, nullptr, 0, nullptr);
        do_codec_emb(e, cx, nullptr, tl.data(), L, nullptr, 1, cl.data(), Cmax + max_length, nullptr, L + Cmax + max_length);
        need = std::max(need, cx.off);
    }
    {
        Ctx cx = make_ctx(B, nullptr, 0, nullptr);
        do_decode(e, cx, nullptr, tl.data(), L, nullptr, cl.data(), Cmax, max_length > 0 ? max_length : 1, 0, 0, 0.f, 0, nullptr, nullptr,
                  nullptr, nullptr, L + 2 + Cmax);
        need = std::max(need, cx.off);
    }
    return need + 4096;
}

int fc_laura_encode(fc_laura* e, const float* text_emb, const int64_t* text_ids, const int32_t* text_lens, int B, int L, float* text_outs,
                    void* workspace, size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    if ((text_emb == nullptr) == (text_ids == nullptr)) return fail("exactly one of text_emb / text_ids");
    if (text_ids && e->arch.vocab_size <= 0) return fail("the checkpoint has no token_embedding");
    if (!text_lens || !text_outs || !workspace || B < 1 || L < 1) return fail("bad argument");
    if (L > e->R) return fail("text longer than max_positions");
    if (check_lens(text_lens, B, 1, L, "text_lens")) return 1;
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    return do_encode(e, cx, text_emb, text_ids, text_lens, L, text_outs);
}

int fc_laura_lm_logprobs(fc_laura* e, const float* text_outs, const int32_t* text_lens, int B, int L, const int64_t* codec,
                         const int32_t* codec_lens, int Cmax, float* logp, int Tseq, void* workspace, size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    if (!text_outs || !text_lens || !logp || !workspace || B < 1 || L < 1) return fail("bad argument");
    if (check_lens(text_lens, B, 1, L, "text_lens")) return 1;
    if (codec && (!codec_lens || check_lens(codec_lens, B, 0, Cmax, "codec_lens"))) return fail("bad codec_lens");
    int need = 0;
    for (int b = 0; b < B; ++b) need = std::max(need, text_lens[b] + 2 + (codec ? codec_lens[b] : 0));
    if (Tseq < need) return fail("Tseq too small");
    if (Tseq > e->R) return fail("sequence longer than max_positions");
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    return do_lm_logprobs(e, cx, text_outs, text_lens, L, codec, codec_lens, Cmax, logp, Tseq);
}

int fc_laura_decode_codec(fc_laura* e, const float* text_outs, const int32_t* text_lens, int B, int L, const int64_t* continual,
                          const int32_t* cont_lens, int Cmax, int max_length, int sampling_mode, int sampling_k, float sampling_p,
                          uint64_t seed, const int64_t* forced, int64_t* tokens, int32_t* out_lens, float* step_logp, void* workspace,
                          size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    if (!text_outs || !text_lens || !tokens || !out_lens || !workspace || B < 1 || B > 16 || L < 1 || max_length < 1)
        return fail("bad argument (1 <= B <= 16 utterances per call)");
    if (sampling_mode < 0 || sampling_mode > 3) return fail("sampling_mode must be 0..3");
    if (sampling_mode == 2 && sampling_k < 1) return fail("top-k sampling needs sampling_k >= 1");
    if (sampling_mode == 3 && !(sampling_p > 0.f)) return fail("nucleus sampling needs sampling_p > 0");
    if (check_lens(text_lens, B, 1, L, "text_lens")) return 1;
    if (continual && (!cont_lens || check_lens(cont_lens, B, 0, Cmax, "cont_lens"))) return fail("bad cont_lens");
    int prefix = 0;
    for (int b = 0; b < B; ++b) prefix = std::max(prefix, text_lens[b] + 2 + (continual ? cont_lens[b] : 0));
    if (prefix + max_length > e->R) return fail("prefix + max_length exceeds max_positions");
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    int rc = do_decode(e, cx, text_outs, text_lens, L, continual, cont_lens, Cmax < 0 ? 0 : Cmax, max_length, sampling_mode, sampling_k,
                       sampling_p, seed, forced, tokens, out_lens, step_logp, prefix);
    if (rc == 2) {
        // A hand-off of the persistent decoding step timed out (its workgroups were not all resident: another stream or process held CUs --
        // the launch is not cooperative; or a lost store).  Bounded spins, the launch has ended, nothing of its output is used: THIS call is
        // run again from the start on the kernel chain (a generation is a function of its seed, so the result is the one the chain gives),
        // later calls stay on the chain until fc_laura_set_persistent_step(e, 1).  The event is counted, not hidden.
        e->persist_fallbacks++;
        Ctx cx2 = make_ctx(B, workspace, workspace_bytes, stream);
        rc = do_decode(e, cx2, text_outs, text_lens, L, continual, cont_lens, Cmax < 0 ? 0 : Cmax, max_length, sampling_mode, sampling_k,
                       sampling_p, seed, forced, tokens, out_lens, step_logp, prefix);
        if (rc == 2) return fail("decode_codec: the decoding step timed out on the kernel chain as well");
    }
    return rc;
}

int fc_laura_persistent_step_fallbacks(const fc_laura* e) { return e ? e->persist_fallbacks : -1; }

int fc_laura_codec_emb(fc_laura* e, const float* text_outs, const int32_t* text_lens, int B, int L, const int64_t* codec, int nq_cols,
                       const int32_t* codec_lens, int Cmax, float* emb, void* workspace, size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    if (!text_outs || !text_lens || !codec || !codec_lens || !emb || !workspace || B < 1 || L < 1 || Cmax < 1) return fail("bad argument");
    if (nq_cols < e->arch.predict_nq) return fail("codec has fewer columns than predict_nq");
    if (check_lens(text_lens, B, 1, L, "text_lens") || check_lens(codec_lens, B, 0, Cmax, "codec_lens")) return 1;
    int Tmax = 0;
    for (int b = 0; b < B; ++b) Tmax = std::max(Tmax, text_lens[b] + codec_lens[b]);
    if (Tmax > e->R) return fail("sequence longer than max_positions");
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    return do_codec_emb(e, cx, text_outs, text_lens, L, codec, nq_cols, codec_lens, Cmax, emb, Tmax);
}

int fc_laura_set_persistent_step(fc_laura* e, int on) {
    if (!e) return -1;
    e->persist_on = on != 0;
    return e->persist_on && e->persist_grid > 0 ? 1 : 0;
}

int fc_laura_debug_probe(void* dev_dst, size_t cap_bytes, int stack, int layer, int what) {
    g_probe.dst = (float*)dev_dst; g_probe.cap = cap_bytes; g_probe.stack = stack; g_probe.layer = layer; g_probe.what = what;
    return 0;
}

int fc_laura_linear(fc_laura* e, const char* name, const float* x, int B, int T, int step_form, float* y, void* workspace,
                    size_t workspace_bytes, void* stream) {
    if (check_ready(e)) return 1;
    if (!name || !x || !y || !workspace || B < 1 || T < 1) return fail("bad argument");
    auto it = e->lin_by_name.find(name);
    if (it == e->lin_by_name.end()) return fail(std::string("unknown Linear: ") + name);
    const Lin& L = *it->second;
    Ctx cx = make_ctx(B, workspace, workspace_bytes, stream);
    if (step_form) {
        if (!L.wf) return fail(std::string(name) + " has no decoding-step form");
        if (B * T > 16) return fail("step form: at most 16 rows");
        cx.check(step_gemv(L, x, B * T, nullptr, nullptr, 0.f, 0, 0, y, L.cout, cx.st), "GEMV");
        return cx.err;
    }
    const int Tp = pad4(T);
    float* in = cx.alloc<float>((size_t)B * L.cin * Tp);
    float* o = cx.alloc<float>((size_t)B * L.cout * Tp);
    if (!cx.live()) return 1;
    cx.check(lk::launch_tm_to_fm(x, nullptr, B, T, L.cin, Tp, in, cx.st), "transpose");
    cx.check(launch_lin_full(L, in, B, Tp, o, cx.st), "GEMM");
    cx.check(lk::launch_fm_to_tm(o, nullptr, B, L.cout, Tp, T, y, cx.st), "transpose");
    return cx.err;
}

}  // extern "C"
