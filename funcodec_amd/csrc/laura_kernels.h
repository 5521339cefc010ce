// Launch interface between the LauraTTS engine (laura.hip) and its gfx950 kernels (laura_kernels.hip).
//
// Two execution forms of the same rel-pos self-attention stacks (funcodec/models/encoder/conformer_encoder.py,
// transformer_encoder.py, funcodec/modules/attention.py:212-308):
//   * FULL-SEQUENCE form (text encoder, LM prefix / teacher forcing, fine codec predictor): activations are FEATURE-MAJOR
//     [B][C][T] (time contiguous), so every Linear is a k = 1 layer of the codec's implicit-GEMM conv kernel (fp32-input MFMA,
//     conv_kernel.h) and attention reads Q / K / V rows as [d][t] slabs;
//   * STEP form (autoregressive decoding with a KV cache): B <= 16 new tokens per step, TOKEN-MAJOR vectors [B][C]; every
//     Linear is a weight-streaming GEMV on 16x16x4 MFMAs (weights in fragment order, one contiguous 1 KiB per wave load),
//     LayerNorm / activation / residual / cache scatter fused into its prologue / epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fc {
namespace laura {

// ---- full-sequence form -------------------------------------------------------------------------------------------------
// y = [relu](LayerNorm_C(x [+ add])) * post_scale;  sum_out (may alias x) receives x + add.  x, add, y: [B][C][T]
hipError_t launch_layernorm_fm(const float* x, const float* add, float* sum_out, const float* gamma, const float* beta, float eps,
                               int relu, float post_scale, float* y, int B, int C, int T, hipStream_t st);
// in-place activation over n floats: 1 = relu, 2 = swish (x * sigmoid(x))
hipError_t launch_act(float* x, size_t n, int act, hipStream_t st);

struct AttnFull {
    const float* qkv = nullptr;     // [B][3 d][T]: rows [0, d) q, [d, 2d) k, [2d, 3d) v
    const float* ptab = nullptr;    // [d][PR]: linear_pos(pe(r)) at column r + R - 1
    const float* bias_u = nullptr;  // [d] (= [h][dk])
    const float* bias_v = nullptr;
    const int* lens = nullptr;      // device [B]: keys >= lens[b] are masked
    const int* bidir = nullptr;     // device [B] or null: with causal, rows and columns < bidir[b] see each other
    int causal = 0;
    float* ctx = nullptr;           // [B][d][T]
    int B = 0, H = 0, DK = 64, T = 0, R = 0, PR = 0;
};
size_t attn_full_lds_bytes(int T);
hipError_t launch_attn_full(const AttnFull& a, hipStream_t st);

// [B][L][D] token-major (rows t >= lens[b] ignored) -> [B][D][T] feature-major, zero for t >= lens[b]
hipError_t launch_tm_to_fm(const float* in, const int* lens, int B, int L, int D, int T, float* out, hipStream_t st);
// [B][D][T] -> [B][L][D], rows t >= lens[b] zero
hipError_t launch_fm_to_tm(const float* in, const int* lens, int B, int D, int T, int L, float* out, hipStream_t st);
// token ids [B][L] (i64, < 0 = padding) -> table rows, feature-major [B][D][T]
hipError_t launch_token_embed_fm(const int64_t* ids, const float* table, int vocab, int B, int L, int D, int T, float* out, hipStream_t st);

// LM input (LauraGenModel.build_llm_io, laura_model.py:204-247): [sos, text, task_id, codec embeddings] per utterance.
//   text_fm [B][D][Tt]; lm_emb [2][D]; cb [nq_all][K][D]; codec [B][Cmax][nq] (i64) or null; out [B][D][T]; seq_len[b] written
hipError_t launch_lm_assemble(const float* text_fm, int Tt, const int* text_lens, const float* lm_emb, const float* cb, int K, int nq,
                              const int64_t* codec, const int* codec_lens, int Cmax, int B, int D, int T, float* out, int* seq_len,
                              int* bidir_len, hipStream_t st);
// fine-predictor input (cal_codec_emb, laura_model.py:296-322): [text, sum of the first nq codebook rows]; split: each part gets
// x * sqrt(D) + pe_abs[position within the part]
hipError_t launch_nar_assemble(const float* text_fm, int Tt, const int* text_lens, const float* cb, int K, int nq, const int64_t* codec,
                               const int* codec_lens, int Cmax, int codec_stride_nq, const float* pe_abs, int split, int B, int D, int T,
                               float* out, int* seq_len, hipStream_t st);
// out [B][Cmax][D] token-major = in [B][D][T] columns text_lens[b] .. text_lens[b] + codec_lens[b] - 1 (zero rows past it)
hipError_t launch_nar_extract(const float* in, const int* text_lens, const int* codec_lens, int B, int D, int T, int Cmax, float* out,
                              hipStream_t st);
// log-softmax over the V rows of every column: in [B][V][T] -> out [B][L][V] (rows t >= lens[b] zero)
hipError_t launch_logsoftmax_fm(const float* in, const int* lens, int B, int V, int T, int L, float* out, hipStream_t st);
// K / V rows of a full-sequence QKV buffer into the step form's caches: kc [B][d][Tcap] (feature-major), vc [B][Tcap][d]
hipError_t launch_kv_store(const float* qkv, const int* lens, int B, int d, int T, int Tcap, float* kc, float* vc, hipStream_t st);
// x_last[b][c] = x[b][c][lens[b] - 1]   ([16][C] token-major, rows >= B untouched)
hipError_t launch_gather_last(const float* x, const int* lens, int B, int C, int T, float* out, hipStream_t st);

// ---- step form ------------------------------------------------------------------------------------------------------------
struct Gemv {
    const float* x = nullptr;       // [B][K] token-major
    const float* wf = nullptr;      // fragment order [ceil(N/16)][K/16][64 lanes][4]
    const float* bias = nullptr;    // [N]
    const float *gamma = nullptr, *beta = nullptr;   // LayerNorm over K applied to x first (null: none)
    float eps = 0.f;
    int act = 0;                    // 0 none, 1 relu, 2 swish
    int mode = 0;                   // 0: y[b][n] = v   1: y[b][n] += v   2: q / K-cache / V-cache scatter
    float* y = nullptr;             // [B][ldy]
    int ldy = 0;
    // mode 2: n < d -> y[b][n];  d <= n < 2d -> kc[b][n - d][pos[b]];  else vc[b][pos[b]][n - 2d]
    float *kc = nullptr, *vc = nullptr;
    const int* pos = nullptr;
    int d = 0, Tcap = 0;
    int B = 0, K = 0, N = 0;
    // x given as the key-range partials of the step attention (combined while staging): [B][H][NS][DK + 2], K = H * DK
    const float* apart = nullptr;
    int H = 0, DK = 0, NS = 0;
};
hipError_t launch_gemv(const Gemv& g, hipStream_t st);
// rows of x [B][C] in place: [relu](LayerNorm(x)) * post_scale
hipError_t launch_layernorm_rows(float* x, const float* gamma, const float* beta, float eps, int relu, float post_scale, int B, int C,
                                 hipStream_t st);

struct AttnStep {
    const float* q = nullptr;       // [B][d]
    const float *kc = nullptr, *vc = nullptr;   // caches of this layer: [B][d][Tcap], [B][Tcap][d]
    const float *ptab = nullptr, *bias_u = nullptr, *bias_v = nullptr;
    const int* pos = nullptr;       // device [B]: position of the query (keys 0 .. pos[b])
    float* part = nullptr;          // [B][H][NS][DK + 2]: per key range the unnormalised context, running max, sum of exponentials
    int NS = 1;                     // key ranges per (utterance, head): B * H * NS workgroups
    int B = 0, H = 0, DK = 64, Tcap = 0, R = 0, PR = 0;
};
hipError_t launch_attn_step(const AttnStep& a, hipStream_t st);

struct Sample {
    const float* logits = nullptr;  // [B][V], V = nq * (K + 1)
    int B = 0, K = 0, nq = 0;
    int mode = 0;                   // 0 greedy, 1 softmax, 2 top-k (ki), 3 nucleus (pf)
    int ki = 0; float pf = 0.f;
    unsigned long long seed = 0;
    const int64_t* forced = nullptr;    // [B][max_steps][nq] or null: teacher forcing (the sampled ids are replaced)
    int max_steps = 0;
    int64_t* tokens = nullptr;      // [B][tok_stride][nq]; generated token g of b goes to row tok_off[b] + g
    int tok_stride = 0;
    const int* tok_off = nullptr;   // device [B]
    int* n_gen = nullptr;           // device [B]: tokens generated so far (incremented here)
    int* done = nullptr;            // device [B]
    int* n_done = nullptr;          // device [1]: number of finished utterances (host polls a copy)
    int* pos = nullptr;             // device [B]: cache fill, incremented for utterances still running
    int* step = nullptr;            // device [B]: samples drawn so far per utterance (RNG stream position), incremented
    float* logp_out = nullptr;      // [B][max_steps][V] or null: log-softmax the step sampled from
    const float* cb = nullptr;      // [nq_all][K][D] codebook table: next LM input = sum_k cb[k][tok_k]
    int D = 0;
    float* next_emb = nullptr;      // [B][D]
    // optional fusion of the LM's input layer for the next step: xs[b] = [relu](LayerNorm_1e-5(W e + bias)) * xscale
    const float *emb_wt = nullptr, *emb_bias = nullptr, *emb_g = nullptr, *emb_b = nullptr;   // emb_wt [D][dm] = W transposed
    int dm = 0, emb_relu = 0;
    float xscale = 1.f;
    float* xs = nullptr;            // [B][dm]
    unsigned* launch_seq = nullptr; // device word incremented once per launch (the persistent step kernel's launch number), or null
};
hipError_t launch_sample(const Sample& s, hipStream_t st);

hipError_t launch_fill_i32(int* p, int v, int n, hipStream_t st);

// ---- step form as ONE persistent launch (laura_persist.hip) ------------------------------------------------------------------------
struct StepLayer {          // constant device pointers of one block (fragment-order weights of the step form, biases padded to row tiles)
    const float *wqkv, *bqkv, *wout, *bout, *wff1, *bff1, *wff2, *bff2;
    const float *n1g, *n1b, *n2g, *n2b, *bu, *bv, *ptab;
};
struct StepPersistArgs {
    const StepLayer* layers;        // device [NL]
    const float *wdec, *bdec;       // output layer (fragment order), bias
    const float *ag, *ab;           // after_norm
    const float* xs;                // [16][d]: LM input of this step (written by the sampler of the previous step)
    float* logits;                  // [16][V]
    float* edge;                    // NL x edge_stride floats: per block q [16][d] | partials [B][H][NS][DK + 4] | xm [16][d] | h [16][ff] | x_out [KS2][16][d]
    float *kc, *vc;                 // KV caches [NL][B][d][Tcap] / [NL][B][Tcap][d]
    const int* pos;                 // device [B]
    unsigned* sync;                 // arrival counters (step_persist_sync_words), zeroed once per decode call
    const unsigned* seq;            // device word: number of this launch within the call (1, 2, ...), incremented by the sampler
    size_t edge_stride;
    int o_q, o_ap, o_xm, o_hb, o_xo;
    int B, d, ff, H, DK, NL, V, Tcap, R, PR, NS, act, G;
    int KS2;                        // k slices of w_2 (step_persist_ksplit)
    int wtile;                      // floats of one LDS weight tile (set by launch_step_persist)
    unsigned long long* trace;      // tuning aid (FC_LAURA_TRACE): [G][64 units][8] s_memtime stamps of the last launch, or null
    int test_timeout;               // test hook (FC_LAURA_PERSIST_TEST=timeout): behave as if a hand-off had timed out
};
int step_persist_ksplit(int d, int ff);
size_t step_persist_sync_words(int NL);
size_t step_persist_lds_bytes(int B, int d, int ff, int H, int DK);
bool step_persist_supported(int B, int d, int ff, int H, int DK, int V, int NS);
int step_persist_grid(int device, int d, int ff);          // workgroups of the persistent launch (0: not launchable here)
hipError_t launch_step_persist(const StepPersistArgs& a, hipStream_t st);

}  // namespace laura
}  // namespace fc
