/*
 * funcodec_amd.h -- C ABI of the MI355X (gfx950) FunCodec encode/decode engine.
 *
 * The reference (modelscope/FunCodec, /root/reference) is 100 % Python on top of torch.nn; it has no
 * FFI layer.  Its operator boundary for this path is the set of Python callables listed next to each
 * entry point below; a maintainer binds this library with ctypes from
 * `funcodec/bin/codec_inference.py` (Speech2Token.__call__, :86-134).
 *
 * Conventions
 *   - plain C types only; every pointer argument marked "dev" is a DEVICE pointer (HBM) owned by the
 *     caller (e.g. `tensor.data_ptr()`), "host" pointers are ordinary host memory;
 *   - all tensors are dense, row-major, fp32 unless stated; code indices are int64 like the reference;
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it (no host sync);
 *   - return value: 0 = ok, non-zero = error, message via fc_last_error() (thread-local);
 *   - one engine per (device, checkpoint); calls on one engine must be serialised by the caller;
 *   - the engine never falls back to a CPU path: without a gfx950 device every compute call fails.
 */
#ifndef FUNCODEC_AMD_H
#define FUNCODEC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FC_MAX_RATIOS 8
#define FC_ABI_VERSION 6     /* layout of fc_arch / fc_laura_arch.  6 (round 4): fc_arch.q0_ds_ratio appended; fc_arch.input_channels = 2 with
                              * model_type 0 is the stereo time-domain codec.  (Round 4 also added entry points that changed no struct:
                              * fc_laura_set_persistent_step, fc_debug_freq_features, fc_q0_source_frames.) */

typedef struct fc_engine fc_engine;

/* Architecture, i.e. the subset of config.yaml the hot path depends on.
 * Replaces: GANSpeechCodecTask.build_model (funcodec/tasks/gan_speech_codec.py:300-343) +
 * SEANetEncoder.__init__ (funcodec/models/encoder/seanet_encoder.py:88-160) +
 * SEANetDecoder.__init__ (funcodec/models/decoder/seanet_decoder.py:88-164) +
 * CostumeQuantizer.__init__ (funcodec/models/quantizer/costume_quantizer.py:7-55). */
typedef struct fc_arch {
    int32_t abi_version;            /* FC_ABI_VERSION */
    int32_t sample_rate;
    int32_t audio_normalize;        /* model_conf.audio_normalize */
    int32_t n_filters;              /* 32 */
    int32_t dimension;              /* 128: encoder output / codebook dim */
    int32_t n_ratios;
    int32_t ratios[FC_MAX_RATIOS];  /* decoder order, e.g. {8,5,4,2,2}; encoder walks it reversed */
    int32_t kernel_size;            /* 7 */
    int32_t last_kernel_size;       /* 7 */
    int32_t residual_kernel_size;   /* 3 */
    int32_t compress;               /* 2 */
    int32_t lstm_layers;            /* 2 (0 = no sequence model) */
    int32_t lstm_skip;              /* res_seq */
    float   elu_alpha;              /* 1.0 */
    float   gn_eps;                 /* 1e-5 */
    int32_t codebook_size;          /* 1024 */
    int32_t num_quantizers;         /* 32 */
    /* conv wrapper flavour (funcodec/modules/normed_modules/conv.py:20-56,205-305), ABI version 2: */
    int32_t norm_type;              /* 0 = GroupNorm(1,C) after every conv ("time_group_norm"); 1 = weight_norm (checkpoint holds
                                       weight_g / weight_v, no output norm); 2 = none (plain weight, no output norm) */
    int32_t causal;                 /* 1: all conv padding on the left, transposed convs trimmed on the right only */
    int32_t n_residual_layers;      /* residual blocks per stage (1 in the encodec recipes, 3 in the SoundStream recipe) */
    int32_t dilation_base;          /* block j of a stage dilates its k=3 conv by dilation_base**j (seanet_encoder.py:127-133) */
    /* ABI version 4: the STFT-domain codec (FreqCodec, funcodec/models/codec_freq.py:123-210; SEANetEncoder2d / SEANetDecoder2d,
     * funcodec/models/encoder/seanet_encoder.py:252-363, decoder/seanet_decoder.py:244-360).  model_type 0 ignores the rest except input_channels. */
    int32_t model_type;             /* 0 = encodec (time domain), 1 = freq_codec (codec_domain [mag_phase, mag_phase] or [mag_angle, mag_angle]) */
    int32_t input_channels;         /* model_type 0: audio channels, 1 (0 reads as 1) or 2 = stereo (config input_size / decoder_conf.channels;
                                     * codec_basic.py:342-344,366: the volume scale is taken from the channel mean); wav buffers are [B][C][T].
                                     * model_type 1: encoder input / decoder output channels of the 2-D nets, which also names the codec_domain like the
                                     * reference's input_size does (codec_freq.py:356-379): 3 = mag_phase (log-magnitude, phase re, phase im),
                                     * 2 = mag_angle (log-magnitude, torch.angle) */
    int32_t n_fft;                  /* 512  (model_conf.domain_conf.n_fft) */
    int32_t stft_hop;               /* 160  (model_conf.domain_conf.hop_length) */
    int32_t ratios_f[FC_MAX_RATIOS];/* frequency ratios of the 2-D stages, decoder order (ratios[] holds the time ratios) */
    /* grouped 2-D convs (seanet_encoder.py:224,234,321; seanet_decoder.py:219,229,324): <= 0 = dense (the recipe), else the layer has
     * groups = min(in, out) / 2 / ratio (res-block convs), channels / 2 / ratio (shortcut, strided / transposed convs) */
    int32_t enc_conv_group_ratio;   /* encoder_conf.conv_group_ratio */
    int32_t dec_conv_group_ratio;   /* decoder_conf.conv_group_ratio */
    int32_t dec_tr_conv_group_ratio;/* decoder_conf.tr_conv_group_ratio */
    /* ABI version 5: CostumeQuantizer's optional projection / range (funcodec/models/quantizer/costume_quantizer.py:23-35,63-73,84-87) */
    int32_t codec_dim;              /* quantizer_conf.codec_dim: 0 (or = dimension) = none; else the codebooks live in codec_dim dims behind
                                       input_proj / output_proj Linears (checkpoint keys quantizer.input_proj.*, quantizer.output_proj.*) */
    float   codec_range;            /* quantizer_conf.codec_range: 0 = none; else the quantiser input is tanh(x) * codec_range */
    /* ABI version 6 */
    int32_t q0_ds_ratio;            /* quantizer_conf.q0_ds_ratio (funcodec/modules/quantization/ddp_core_vq.py:354-356,396-404): <= 1 = off;
                                       > 1: the FIRST quantiser stage sees the nearest-neighbour half-rate sequence (the reference halves
                                       whatever the value is) and its output / indices are repeated back to Tf frames */
} fc_arch;

/* ---- lifetime ------------------------------------------------------------------------------------ */
int  fc_abi_version(void);
const char* fc_last_error(void);

/* Replaces build_model(): builds the layer plan; allocates nothing on the device yet. */
int  fc_engine_create(const fc_arch* arch, int device, fc_engine** out);
void fc_engine_destroy(fc_engine* e);

/* Checkpoint contract.  Replaces build_model_from_file / filter_state_dict
 * (funcodec/tasks/abs_task.py:1895-1947, funcodec/torch_utils/load_pretrained_model.py:12-43):
 * the host enumerates the state_dict keys the engine wants and hands each tensor over by name. */
int  fc_engine_num_weights(const fc_engine* e);
/* name/dims of expected tensor i (dims has room for 4 entries); returns ndim or <0. */
int  fc_engine_weight_info(const fc_engine* e, int i, const char** name, int64_t* dims);
/* host fp32 tensor in the reference's own layout (e.g. Conv1d [Cout,Cin,k], ConvTranspose1d [Cin,Cout,k],
 * LSTM weight_ih_l0 [4H,H], quantizer.rq.model.embed [n_q,K,D]).  Unknown names -> error code 2
 * (the host skips discriminator.* etc. itself).  Shape mismatch -> error. */
int  fc_engine_set_weight(fc_engine* e, const char* name, const float* host, const int64_t* dims, int ndim);
/* Folds / re-lays-out the weights into HBM.  Fails if a tensor is missing. */
int  fc_engine_finalize(fc_engine* e);

/* ---- sizes --------------------------------------------------------------------------------------- */
int    fc_engine_hop_length(const fc_engine* e);
/* frames emitted for n_samples: ceil at every encoder stride (SConv1d extra padding, conv.py:57-64). */
int    fc_engine_frames(const fc_engine* e, int n_samples);
/* samples the decoder emits for n_frames frames: n_frames * hop for the time-domain codec; stft_hop * (frames * time ratios - 1) for
 * the STFT-domain codec (torch.istft with center=True).  Upper bound of `out_len` of the decode entry points. */
int    fc_engine_decoded_samples(const fc_engine* e, int n_frames);
/* bytes of caller-provided device scratch needed by any call with batch B and T samples (or Tf*hop).  The figure INCLUDES 4 KiB of tail
 * slack that every entry point requires behind its last internal buffer (kernels with unclamped row-end loads and the DMA-staged conv
 * read a few bytes past a buffer's end): a workspace that ends exactly at the last buffer is refused ("workspace too small"), never
 * over-read.  The workspace pointer itself must be 256-byte aligned device memory. */
size_t fc_engine_workspace_bytes(const fc_engine* e, int B, int T);

/* ---- the hot path -------------------------------------------------------------------------------- */
/* Encodec.inference_encoding (funcodec/models/codec_basic.py:720-764) = _encode_frame (:361-380) +
 * SEANetEncoder.forward + CostumeQuantizer.inference -> DRVQ.forward (ddp_core_vq.py:367-418).
 *   wav        dev f32 [B,T]       ([B,2,T] for a stereo model, fc_arch.input_channels = 2 with model_type 0; B and T keep their meaning)
 *   n_q        number of quantizers to run (1..num_quantizers)
 *   codes      dev i64 [n_q,B,Tf]                         (code_indices[0])
 *   quantized  dev f32 [B,Tf,D]   or NULL                 (code_embeddings[0][0])
 *   sub_quants dev f32 [n_q,B,D,Tf] or NULL               (sub_quants[0])
 *   scale      dev f32 [B]        or NULL (1e-8+rms; written only if audio_normalize)
 *   enc_out    dev f32 [B,Tf,D]   or NULL (encoder output before quantisation) */
int fc_encode(fc_engine* e, const float* wav, int B, int T, int n_q,
              int64_t* codes, float* quantized, float* sub_quants, float* scale, float* enc_out,
              void* workspace, size_t workspace_bytes, void* stream);

/* Encodec.inference_decoding_emb (codec_basic.py:804-836) = _decode_frame (:398-408) + SEANetDecoder.forward.
 *   emb   dev f32 [B,Tf,D];  scale dev f32 [B] or NULL (multiplied in when non-NULL, :406-407)
 *   wav   dev f32 [B,out_len] ([B,2,out_len] for a stereo model), out_len <= fc_engine_decoded_samples(Tf) (the first out_len samples
 *         of every channel are written) */
int fc_decode_emb(fc_engine* e, const float* emb, const float* scale, int B, int Tf, int out_len,
                  float* wav, void* workspace, size_t workspace_bytes, void* stream);

/* Encodec.inference_decoding (codec_basic.py:766-802): DRVQ.decode (ddp_core_vq.py:442-453) + decoder.
 *   codes dev i64 [B,Tf,n_q] (the reference's token layout);  emb_out dev f32 [B,Tf,D] or NULL */
int fc_decode_codes(fc_engine* e, const int64_t* codes, int B, int Tf, int n_q, int out_len,
                    float* wav, float* emb_out, void* workspace, size_t workspace_bytes, void* stream);

/* Encodec.inference (codec_basic.py:670-718) / FreqCodec.inference (codec_freq.py): encode + decode in one enqueue;
 * recon [B, min(T, fc_engine_decoded_samples(frames))] (= the reference's recon[:, :, :T]; always T for model_type 0; [B,2,T] for a
 * stereo model). */
int fc_encode_decode(fc_engine* e, const float* wav, int B, int T, int n_q, int use_scale,
                     int64_t* codes, float* quantized, float* sub_quants, float* scale, float* recon,
                     void* workspace, size_t workspace_bytes, void* stream);

/* _linear_overlap_add (funcodec/models/codec_basic.py:77-116), the tail of Encodec._decode (:382-396) when
 * model_conf.segment_dur is set: triangle-weighted overlap-add of the decoded segments, products accumulated in
 * frame order and divided once by the summed weights, exactly as the reference orders it.
 *   frames      dev array of n_frames dev pointers, frame f = f32 [B, lens[f]] (decoded, UNTRIMMED segment f, which
 *               starts at sample f*stride); lens dev i32 [n_frames]; frame0_len = lens[0] (sizes the window, host copy)
 *   out         dev f32 [B, out_len]: the first out_len samples of the sum (Encodec.inference trims to the input, :711)
 * Rows are independent: a stereo model passes B * 2 rows. */
int fc_overlap_add(const float* const* frames, const int* lens, int n_frames, int B, int frame0_len, int stride,
                   int out_len, float* out, void* stream);

/* Deferred device-side failures.  Kernels cannot return a status, so two conditions are recorded in host-visible
 * status words and reported by the NEXT fc_* compute call on the engine (non-zero return, message in fc_last_error(),
 * condition cleared) or by this call.  *flags (may be NULL) receives the conditions pending at entry:
 *   FC_STATUS_FLAG_LSTM_TIMEOUT  the persistent LSTM kernel's grid barrier timed out (workgroups not co-resident);
 *                                that call's outputs are NaN-poisoned; the engine falls back to per-step launches
 *   FC_STATUS_FLAG_BAD_CODE      fc_decode_codes saw an index outside [0, codebook_size) (F.embedding would raise)
 * Call it after synchronising the stream to learn about the calls enqueued so far. */
#define FC_STATUS_FLAG_LSTM_TIMEOUT 1u
#define FC_STATUS_FLAG_BAD_CODE 2u
int fc_engine_status(fc_engine* e, unsigned* flags);

/* ---- per-op entry points (so tests can pin each kernel against torch.nn.functional) -------------- */
/* DRVQ.forward on rows: x dev f32 [N,D], codebooks as loaded; codes dev i64 [n_q,N];
 * quantized dev f32 [N,D] or NULL.  With fc_arch.q0_ds_ratio > 1 the rows are ONE utterance of N >= 2 frames and
 * `workspace` must hold N * 4 bytes (the stage-0 source-row table); otherwise the workspace is unused. */
int fc_rvq_encode(fc_engine* e, const float* x, int N, int n_q, int64_t* codes, float* quantized,
                  void* workspace, size_t workspace_bytes, void* stream);

/* HOST function, no GPU: frames[t] = the frame whose stage-0 code frame t receives when quantizer_conf.q0_ds_ratio > 1, i.e. the
 * composition of the reference's two nearest-neighbour F.interpolate calls (ddp_core_vq.py:396-404: Tf -> Tf // 2 -> Tf) that the
 * kernels apply as a row table.  Exported so that the restatement of torch's index arithmetic is tested against torch itself. */
int fc_q0_source_frames(int Tf, int32_t* frames /* host, Tf entries */);

/* One SConv1d / SConvTranspose1d of the plan, addressed by its checkpoint prefix (e.g.
 * "encoder.model.3.conv", "decoder.model.3.convtr"): y = GroupNorm(conv(pad(act(x)))) as the reference
 * module computes it (conv.py:243-305), x dev f32 [B,Cin,T], y dev f32 [B,Cout,Tout] (trimmed for convtr).
 * apply_elu: apply ELU to x first (the nn.ELU that precedes the module in the Sequential). */
int fc_layer_forward(fc_engine* e, const char* prefix, const float* x, int B, int T, int apply_elu,
                     float* y, void* workspace, size_t workspace_bytes, void* stream);
/* output length of that layer for input length T */
int fc_layer_out_len(const fc_engine* e, const char* prefix, int T);

/* SEANetResnetBlock.forward (seanet_encoder.py:44-61; decoder copy seanet_decoder.py:42-59) addressed by its Sequential
 * prefix ("encoder.model.1", "decoder.model.16"): y = shortcut(x) + block(x), each conv followed by its GroupNorm (when the
 * recipe has one); x, y dev f32 [B,C,T].  Exercises the fused shortcut + block.1 launch of the thin (C <= 64) blocks. */
int fc_resblock_forward(fc_engine* e, const char* prefix, const float* x, int B, int T,
                        float* y, void* workspace, size_t workspace_bytes, void* stream);

/* SLSTM.forward (lstm.py:22-28) addressed by prefix ("encoder.model.16.lstm"): x,y dev f32 [B,C,T]. */
int fc_lstm_forward(fc_engine* e, const char* prefix, const float* x, int B, int T,
                    float* y, void* workspace, size_t workspace_bytes, void* stream);

/* ---- profiling aid ------------------------------------------------------------------------------- */
/* Algorithmic work of one fc_encode_decode call: flops and bytes, total and for the
 * implicit-GEMM conv kernel family only. */
typedef struct fc_work {
    double total_flops, total_bytes;
    double conv_flops, conv_bytes;
    double lstm_flops, rvq_flops;
    int32_t conv_launches, total_launches;
} fc_work;
int fc_engine_work(const fc_engine* e, int B, int T, int n_q, fc_work* out);

/* Optional in-engine timing: when enabled, every conv launch (and each LSTM block / RVQ launch) is
 * bracketed by hipEventRecord on the caller's stream.  fc_engine_profile_read() synchronises on the
 * last event, returns per-kernel-class totals accumulated since the last read and resets them. */
typedef struct fc_prof {
    char    kernel[64];      /* named like rocprofv3 prints it, e.g. "conv_mfma_kernel<128, 128, 2, 2, 0, 8, false>",
                                "lstm_persist_kernel<NS> (...)" for the persistent recurrence */
    double  total_ms;        /* sum of event-to-event durations */
    double  flops, bytes;    /* algorithmic work of those launches */
    int32_t launches;
    int32_t reserved;
} fc_prof;
#define FC_PROF_CLASSES 48
int fc_engine_profile(fc_engine* e, int enable);
/* fills out[0..n) (n <= FC_PROF_CLASSES, returned through *n_out); unused entries have launches == 0 */
int fc_engine_profile_read(fc_engine* e, fc_prof* out /* [FC_PROF_CLASSES] */);

/* Kernel-phase timeline of the last conv launch, [2 roles][24 work items][8 stamps] of shader-clock ticks.
 * Only builds made with FC_TIMELINE=1 record anything (all zeros otherwise); a tuning aid, not part of the path. */
int fc_debug_timeline(unsigned long long* dst /* [2*24*8] */);

/* Host-side description of the implicit-GEMM conv kernel's operand layout for one chunk shape (k taps, CC channels per chunk, BM x BN tile);
 * no GPU work -- what the CPU tests check the weight packing and the B-operand offset table against.
 *   info[0] 1 = quad-k layout   info[1] floats per packed weight chunk   info[2] entries of the offset table   info[3] slab row stride
 *   info[4] columns per stride phase (PL)   info[5] slab width in input columns
 *   pack_index (optional) [k][CC][BM]: float index of W[row][chunk channel][tap] inside the packed chunk image
 *   koff (optional): the B-operand offset table (floats), one entry per half-quad (quad layout) or per k-step (round-4 layout) */
int fc_debug_conv_layout(int k, int stride, int dil, int CC, int BM, int BN, int row, int* info /* [6] */, int* pack_index, size_t pack_cap,
                         int* koff, size_t koff_cap);

/* Test hook for the STFT-domain codec (model_type 1): the NEXT fc_encode / fc_encode_decode call of this thread hands its feature tensor
 * (the 2-D encoder's input, codec_freq.py:356-379) to `dev_buf` (mode 1) or takes it from there (mode 2), in the reference's layout
 * [B][input_channels][n_fft / 2 + 1][1 + T / stft_hop] fp32; mode 0 disarms.  One shot.  Why it exists: torch.angle of a bin whose
 * imaginary part is rounding noise around a negative real part is +pi or -pi by the FFT's rounding, so for codec_domain mag_angle no
 * second STFT implementation reproduces the reference's feature tensor bin for bin; the parity tests compare the features modulo 2 pi
 * and pin the rest of the path from the reference's own features. */
int fc_debug_freq_features(void* dev_buf, size_t cap_bytes, int mode);

/* ---- host-side wire formats of the CLI (no GPU work) -------------------------------------------
 * The text form of one utterance's codes, byte for byte what funcodec/bin/codec_inference.py:295-299 writes with
 * json.dumps(indices[:, b, :len].tolist()) for the single frame of non-segmented inference: "[[[i, i, ...], [...], ...]]"
 * (n_q rows of `len` integers, ", " separators).
 *   codes   HOST i64 [n_q][B][T];  out: at least fc_codec_json_bound(n_q, len) bytes;  *written = bytes produced (no terminator) */
size_t fc_codec_json_bound(int n_q, int len);
int fc_format_codec_json(const int64_t* codes, int n_q, int B, int T, int b, int len, char* out, size_t cap, size_t* written);
/* save_audio (codec_inference.py:153-161): peak-rescale to 0.99 (rescale != 0; else clamp to +-0.99), round(x * 32768) clamped to
 * int16, mono 16-bit PCM RIFF file.   wav HOST f32 [n] */
int fc_write_wav_pcm16(const char* path, const float* wav, int n, int sample_rate, int rescale);

/* ======================================================================================================================
 * LauraTTS generation (ABI version 5; SURVEY.md §8f rank 3, BASELINE.json configs[4]): text -> conformer text encoder ->
 * decoder-only rel-pos transformer LM sampling the first `predict_nq` codec groups autoregressively -> non-autoregressive
 * conformer predicting the dense codec embedding -> fc_decode_emb of the codec engine above.
 * Replaces, for inference, funcodec/models/audio_generation/laura_model.py (LauraGenModel.encode :186-202, decode_codec :501-548,
 * cal_codec_emb :296-333 as syn_audio :550-567 calls it), funcodec/lm/transformer_lm.py (TransformerEmbedLM.score :266-313),
 * funcodec/models/encoder/conformer_encoder.py / transformer_encoder.py (the three rel-pos stacks) and
 * funcodec/modules/attention.py:212-308.  The reference re-scores the whole prefix for every token (no KV cache, batch 1, one
 * host round trip per token); this engine keeps a KV cache, decodes a batch of <= 16 prompts per call and samples on the device.
 * Same conventions as above: "dev" = device pointer, "host" = host pointer, fp32, int64 token ids, work enqueued on `stream`
 * (fc_laura_decode_codec additionally synchronises the stream, see there). */
typedef struct fc_laura fc_laura;

/* one rel-pos self-attention stack: ConformerEncoder without CNN / macaron modules (conformer_encoder.py:317-532) or
 * TransformerEncoder_s0 (transformer_encoder.py:424-654) */
typedef struct fc_laura_stack {
    int32_t idim, d_model, heads, ff, layers;
    int32_t act;          /* FFN activation: 1 = ReLU (TransformerEncoder_s0), 2 = Swish (conformer) */
    int32_t embed_relu;   /* ReLU after the input layer's LayerNorm (transformer_encoder.py:463-469) */
    int32_t norm_style;   /* state_dict names of the block norms: 0 = norm_mha / norm_ff, 1 = norm1 / norm2 */
} fc_laura_stack;

typedef struct fc_laura_arch {
    int32_t abi_version;            /* FC_ABI_VERSION */
    int32_t input_size;             /* width of the text embeddings (T5: 1536) or of token_embedding */
    int32_t vocab_size;             /* > 0: the checkpoint holds token_embedding [vocab_size][input_size] */
    int32_t codebook_size;          /* 1024 (the reference's index shift is hard-wired to it, laura_model.py:29) */
    int32_t codebook_dim;           /* 128 */
    int32_t num_quantizers;         /* rows of quantizer_codebook.embed */
    int32_t predict_nq;             /* codec groups the LM predicts per frame */
    int32_t pos_emb_split;          /* model_conf.pos_emb_type: 1 = "split" (abs. positional encoding per part, laura_model.py:312-317), 0 = "uni" */
    int32_t bidirectional_inputs;   /* codec_lm_conf.bidirectional_inputs (transformer_lm.py:286-288) */
    int32_t max_positions;          /* longest sequence any stack will see (sizes the relative-position tables; <= 2048) */
    fc_laura_stack text_encoder, codec_lm, codec_encoder;
} fc_laura_arch;

/* Text2AudioGenTask.build_model (funcodec/tasks/text2audio_generation.py:202-247) */
int  fc_laura_create(const fc_laura_arch* arch, int device, fc_laura** out);
void fc_laura_destroy(fc_laura* e);
/* checkpoint contract, as fc_engine_*: state_dict names of LauraGenModel (host fp32 tensors in the reference's layout) */
int  fc_laura_num_weights(const fc_laura* e);
int  fc_laura_weight_info(const fc_laura* e, int i, const char** name, int64_t* dims);
int  fc_laura_set_weight(fc_laura* e, const char* name, const float* host, const int64_t* dims, int ndim);
int  fc_laura_finalize(fc_laura* e);
/* device scratch for any call with B utterances, texts of <= L tokens, <= Cmax prompt / codec tokens and max_length new tokens */
size_t fc_laura_workspace_bytes(const fc_laura* e, int B, int L, int Cmax, int max_length);

/* LauraGenModel.encode (laura_model.py:186-202): text encoder + text_enc_out_layer.
 *   text_emb  dev f32 [B][L][input_size] or NULL;  text_ids dev i64 [B][L] or NULL (token_embedding lookup,
 *             bin/text2audio_inference.py:99-113; ids < 0 = padding): exactly one of the two
 *   text_lens host i32 [B];   text_outs dev f32 [B][L][codebook_dim] (rows >= text_lens[b] zero) */
int fc_laura_encode(fc_laura* e, const float* text_emb, const int64_t* text_ids, const int32_t* text_lens, int B, int L,
                    float* text_outs, void* workspace, size_t workspace_bytes, void* stream);

/* TransformerEmbedLM.score (transformer_lm.py:266-313) at EVERY position of [<sos>, text, <task>, codec...] in one pass
 * (teacher forcing): logp[b][t] = log_softmax of the decoder output at position t, i.e. what decode_codec samples token
 * t - text_lens[b] - 1 from.   codec dev i64 [B][Cmax][predict_nq] or NULL, codec_lens host i32 [B] or NULL;
 * logp dev f32 [B][Tseq][vocab], Tseq >= max_b(text_lens[b] + 2 + codec_lens[b]), vocab = predict_nq * (codebook_size + 1) */
int fc_laura_lm_logprobs(fc_laura* e, const float* text_outs, const int32_t* text_lens, int B, int L, const int64_t* codec,
                         const int32_t* codec_lens, int Cmax, float* logp, int Tseq, void* workspace, size_t workspace_bytes, void* stream);

/* LauraGenModel.decode_codec (laura_model.py:501-548) for a batch: prefix pass, then one KV-cached step per token, sampled on the device.
 *   continual  dev i64 [B][Cmax][predict_nq] or NULL, cont_lens host i32 [B] or NULL: prompt tokens (zero-shot continuation)
 *   sampling_mode 0 greedy (sampling=False) | 1 softmax (True) | 2 top-k (int, sampling_k) | 3 nucleus (float, sampling_p)
 *   seed       counter-based generator key; the same (seed, inputs) reproduce the same tokens
 *   forced     dev i64 [B][max_length][predict_nq] or NULL: teacher forcing (sampled ids are replaced by these)
 *   tokens     dev i64 [B][Cmax + max_length][predict_nq]: prompt tokens followed by the generated ones (<eos> step dropped)
 *   out_lens   host i32 [B]: valid rows of tokens[b] (filled after an internal stream synchronise, like the reference's .item())
 *   step_logp  dev f32 [B][max_length][vocab] or NULL: the log-probability vector every step sampled from
 * Utterances end at <eos> (any group) or after max_length steps; the call ends when all have. */
int fc_laura_decode_codec(fc_laura* e, const float* text_outs, const int32_t* text_lens, int B, int L, const int64_t* continual,
                          const int32_t* cont_lens, int Cmax, int max_length, int sampling_mode, int sampling_k, float sampling_p,
                          uint64_t seed, const int64_t* forced, int64_t* tokens, int32_t* out_lens, float* step_logp,
                          void* workspace, size_t workspace_bytes, void* stream);

/* LauraGenModel.cal_codec_emb (laura_model.py:296-333) on one-hot probabilities, as syn_audio (:550-567) calls it:
 *   codec dev i64 [B][Cmax][nq_cols] (the first predict_nq columns are used), codec_lens host i32 [B]
 *   emb   dev f32 [B][Cmax][codebook_dim] (rows >= codec_lens[b] zero): the input of fc_decode_emb */
int fc_laura_codec_emb(fc_laura* e, const float* text_outs, const int32_t* text_lens, int B, int L, const int64_t* codec, int nq_cols,
                       const int32_t* codec_lens, int Cmax, float* emb, void* workspace, size_t workspace_bytes, void* stream);

/* per-op entry point (tests pin each Linear against torch.nn.functional.linear): `name` = state_dict prefix of a Linear
 * ("codec_lm.encoder.encoders.3.feed_forward.w_1", "text_enc_out_layer", ...; "<block>.self_attn.linear_qkv" = the fused q/k/v).
 * x dev f32 [B][T][in], y dev f32 [B][T][out].  step_form != 0: through the decoding step's GEMV (LM layers only, B*T <= 16). */
int fc_laura_linear(fc_laura* e, const char* name, const float* x, int B, int T, int step_form, float* y,
                    void* workspace, size_t workspace_bytes, void* stream);

/* debugging aid, not part of the path: the NEXT full-sequence stack runs of this thread copy one intermediate tensor (feature-major
 * [B][rows][T padded to 4]) to dev_dst.  stack 0 text_encoder / 1 codec_lm / 2 codec_encoder (-1 any); what 0 = stream after the input
 * layer, 1 = attention-norm output of block `layer`, 2 = its q/k/v, 3 = its attention context, 5 = residual stream at the entry of
 * block `layer` (layer = number of blocks: before after_norm).  dev_dst NULL switches it off. */
int fc_laura_debug_probe(void* dev_dst, size_t cap_bytes, int stack, int layer, int what);

/* How fc_laura_decode_codec runs a decoding step (no reference counterpart: the reference re-scores the whole prefix per token,
 * funcodec/models/audio_generation/laura_model.py:501-548).  on = 1 (default; FC_LAURA_PERSIST=0 in the environment changes the default):
 * ONE persistent launch per step whose workgroups hand the token vectors to each other through arrival counters (csrc/laura_persist.hip);
 * on = 0: the chain of one kernel per Linear / attention (csrc/laura_kernels.hip).  Same arithmetic, results agree to fp32 rounding.
 * Returns 1 if the persistent form is in effect afterwards, 0 if the chain is (switched off, or the model / device cannot run it), -1 on a null handle. */
int fc_laura_set_persistent_step(fc_laura* e, int on);
/* The persistent launch needs all its workgroups resident; it is not a cooperative launch, so CUs held by another stream or process can make
 * a hand-off time out (bounded spins, no hang).  fc_laura_decode_codec then runs THAT call again on the kernel chain and succeeds with the
 * chain's result (a generation is a function of its seed), and the engine stays on the chain until fc_laura_set_persistent_step(e, 1).
 * This counter says how many calls of this engine went that way (0 normally; -1 on a null handle): a fallback is reported, never silent. */
int fc_laura_persistent_step_fallbacks(const fc_laura* e);

#ifdef __cplusplus
}
#endif
#endif /* FUNCODEC_AMD_H */
