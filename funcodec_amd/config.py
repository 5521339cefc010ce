"""Architecture description of the FunCodec encode/decode hot path, parsed from the reference's
own ``config.yaml`` format.

Mirrors what the reference reads in ``GANSpeechCodecTask.build_model``
(/root/reference/funcodec/tasks/gan_speech_codec.py:300-343) and the constructor defaults of
``SEANetEncoder`` (funcodec/models/encoder/seanet_encoder.py:88-97), ``SEANetDecoder``
(funcodec/models/decoder/seanet_decoder.py:88-96), ``CostumeQuantizer``
(funcodec/models/quantizer/costume_quantizer.py:7-22) and ``Encodec``
(funcodec/models/codec_basic.py).  Only the time-domain ``encodec`` family is
accepted; everything else raises ``NotImplementedError`` naming the unsupported key, so a
checkpoint this engine cannot reproduce is refused instead of silently mis-decoded.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Any, Dict, List, Optional, Tuple


@dataclasses.dataclass
class ArchSpec:
    # audio / model level (Encodec.__init__)
    sample_rate: int = 16000
    input_channels: int = 1
    audio_normalize: bool = True
    # SEANet (shared by encoder and decoder; the recipe gives both the same values)
    n_filters: int = 32
    dimension: int = 128
    ratios: Tuple[int, ...] = (8, 5, 4, 2)       # decoder order; the encoder walks it reversed
    kernel_size: int = 7
    last_kernel_size: int = 7
    residual_kernel_size: int = 3
    n_residual_layers: int = 1
    dilation_base: int = 2
    compress: int = 2
    lstm_layers: int = 2                          # seq_layer_num when seq_model == "lstm", else 0
    lstm_skip: bool = True                        # res_seq
    elu_alpha: float = 1.0
    gn_eps: float = 1e-5
    # quantizer (CostumeQuantizer -> ResidualVectorQuantizer)
    codebook_size: int = 1024
    codebook_dim: int = 128                       # dims the codebooks live in (= codec_dim when the quantiser projects, else dimension)
    codec_range: Optional[float] = None           # CostumeQuantizer: quantiser input = tanh(x) * codec_range (costume_quantizer.py:32-35)
    num_quantizers: int = 32
    encoder_hop_length: int = 320
    quantizer_sampling_rate: int = 16000
    use_ddp: bool = True
    # model_conf.bypass_quantizer (codec_basic.py:148,700-701 / codec_freq.py:698): Encodec.inference returns the ENCODER output as the code
    # embeddings, zero indices [B, Tf] and zero sub_quants, and decodes from the encoder output; inference_encoding still quantises
    bypass_quantizer: bool = False
    # > 1: the first stage quantises the nearest-neighbour half-rate sequence (ddp_core_vq.py:354-356,396-404; the value itself is not
    # used by the reference beyond "> 1": it always halves)
    q0_ds_ratio: int = 1
    # conv wrapper flavour (modules/normed_modules/conv.py): GroupNorm(1, C) after every conv, or weight-normalised convs /
    # plain convs without an output norm; causal = all padding on the left, transposed convs trimmed on the right only
    norm: str = "time_group_norm"
    causal: bool = False
    # Encodec framing (codec_basic.py:288-298): None = one frame; else frames of segment_dur seconds, hop (1-overlap)*length
    segment_dur: Optional[float] = None
    overlap_ratio: float = 0.01
    # STFT-domain codec (model: freq_codec, codec_freq.py:123-210; 2-D SEANet, seanet_encoder.py:252-363): `ratios` then holds
    # the TIME ratios and `ratios_f` the frequency ratios of the stages (decoder order)
    model_type: str = "encodec"                   # "encodec" | "freq_codec" (codec_domain [mag_phase, mag_phase])
    ratios_f: Tuple[int, ...] = ()
    n_fft: int = 512
    stft_hop: int = 160
    # grouped 2-D convs (seanet_encoder.py:224,234,321; seanet_decoder.py:219,229,324): <= 0 = dense
    enc_conv_group_ratio: int = -1
    dec_conv_group_ratio: int = -1
    dec_tr_conv_group_ratio: int = -1

    @property
    def segment_length(self) -> Optional[int]:
        return None if self.segment_dur is None else int(self.segment_dur * self.sample_rate)

    @property
    def segment_stride(self) -> Optional[int]:
        sl = self.segment_length
        return None if sl is None else max(1, int((1 - self.overlap_ratio) * sl))

    @property
    def hop_length(self) -> int:
        return int(math.prod(self.ratios)) * (self.stft_hop if self.model_type == "freq_codec" else 1)

    @property
    def n_stages(self) -> int:
        return len(self.ratios)

    @property
    def bottleneck_channels(self) -> int:
        return self.n_filters * (2 ** len(self.ratios))

    def num_quantizers_for_bandwidth(self, bandwidth: Optional[float]) -> int:
        """funcodec/modules/quantization/vq.py:105-117 (and the implicit ``layers[:n_q]`` cap)."""
        bw_per_q = math.log2(self.codebook_size) * self.quantizer_sampling_rate / self.encoder_hop_length
        n_q = self.num_quantizers
        if bandwidth and bandwidth > 0.0:
            n_q = int(max(1, math.floor(bandwidth / bw_per_q)))
        return min(n_q, self.num_quantizers)

    def frames_for(self, n_samples: int) -> int:
        """Number of codec frames the encoder emits for ``n_samples`` (ceil at every stride)."""
        t = n_samples
        if self.model_type == "freq_codec":
            t = 1 + t // self.stft_hop            # torch.stft, center=True
        for r in reversed(self.ratios):
            t = -(-t // r)
        return t


def _unsupported(key: str, value: Any, why: str = "") -> NotImplementedError:
    return NotImplementedError(
        f"config key {key}={value!r} is outside the MI355X hot-path scope (SURVEY.md §8){': ' + why if why else ''}")


def _q0_ds_ratio(q: Dict[str, Any]) -> int:
    """quantizer_conf.q0_ds_ratio (costume_quantizer.py:19,47 -> vq.py:55,83 -> ddp_core_vq.py:354-356).  Only the distributed quantiser
    implements it: with use_ddp false the reference's constructor raises (core_vq.ResidualVectorQuantization takes no such keyword)."""
    r = q.get("q0_ds_ratio", 1)
    if not isinstance(r, int) or isinstance(r, bool) or r < 1:
        raise _unsupported("quantizer_conf.q0_ds_ratio", r, "a positive integer")
    if r > 1 and not q.get("use_ddp", True):
        raise _unsupported("quantizer_conf.q0_ds_ratio with use_ddp: false", r, "only the distributed quantiser has a first-stage down-sampling")
    return r


def _check_seanet_conf(conf: Dict[str, Any], which: str) -> Dict[str, Any]:
    conf = dict(conf or {})
    norm = conf.get("norm", "weight_norm")
    if norm not in ("time_group_norm", "weight_norm", "none"):
        raise _unsupported(f"{which}.norm", norm, "time_group_norm, weight_norm and none are supported")
    if conf.get("causal", False) and norm == "time_group_norm":
        raise _unsupported(f"{which}.causal", True, "the reference refuses GroupNorm with causal=True (conv.py:46-47)")
    if conf.get("trim_right_ratio", 1.0) != 1.0:
        raise _unsupported(f"{which}.trim_right_ratio", conf["trim_right_ratio"])
    if conf.get("pad_mode", "reflect") != "reflect":
        raise _unsupported(f"{which}.pad_mode", conf["pad_mode"])
    if conf.get("activation", "ELU") != "ELU":
        raise _unsupported(f"{which}.activation", conf["activation"])
    if conf.get("true_skip", False):
        raise _unsupported(f"{which}.true_skip", True)
    if conf.get("add_snake_activation", False):
        raise _unsupported(f"{which}.add_snake_activation", True)
    if not conf.get("double_filters", True) or not conf.get("half_filters", True):
        raise _unsupported(f"{which}.double_filters/half_filters", False)
    if conf.get("final_activation", None) is not None:
        raise _unsupported(f"{which}.final_activation", conf["final_activation"])
    seq_model = conf.get("seq_model", "lstm")
    if seq_model not in ("lstm", None, "none", "None"):
        raise _unsupported(f"{which}.seq_model", seq_model)
    if not 1 <= int(conf.get("n_residual_layers", 1)) <= 8:
        raise _unsupported(f"{which}.n_residual_layers", conf["n_residual_layers"])
    nk = dict(conf.get("norm_params", {}) or {})
    if nk.get("num_groups", 1) != 1:
        raise _unsupported(f"{which}.norm_params.num_groups", nk["num_groups"])
    return conf


def _freq_arch_from_config(cfg: Dict[str, Any]) -> ArchSpec:
    """``model: freq_codec`` (FreqCodec, codec_freq.py:123-210) with the 2-D SEANet nets; the recipes'
    ``codec_domain: [mag_phase, mag_phase]`` and ``[mag_angle, mag_angle]`` are built (GroupNorm or weight_norm nets, the latter optionally
    causal; grouped convs)."""
    if cfg.get("encoder") != "encodec_seanet_encoder_2d" or cfg.get("decoder") != "encodec_seanet_decoder_2d":
        raise _unsupported("encoder/decoder", (cfg.get("encoder"), cfg.get("decoder")), "freq_codec needs the 2-D SEANet nets")
    if cfg.get("quantizer", "costume_quantizer") != "costume_quantizer":
        raise _unsupported("quantizer", cfg.get("quantizer"))
    enc, dec = dict(cfg.get("encoder_conf", {}) or {}), dict(cfg.get("decoder_conf", {}) or {})
    q, m = dict(cfg.get("quantizer_conf", {}) or {}), dict(cfg.get("model_conf", {}) or {})
    for which, conf in (("encoder_conf", enc), ("decoder_conf", dec)):
        nrm, cau = conf.get("norm", "weight_norm"), bool(conf.get("causal", False))
        if nrm not in ("time_group_norm", "weight_norm"):
            raise _unsupported(f"{which}.norm", nrm, "time_group_norm or weight_norm")
        if nrm == "time_group_norm" and cau:
            raise _unsupported(f"{which}.causal", True, "GroupNorm convs cannot be causal (the reference refuses it too, conv.py:46-47)")
        if dict(conf.get("norm_params", {}) or {}).get("num_groups", 1) != 1:
            raise _unsupported(f"{which}.norm_params.num_groups", conf["norm_params"]["num_groups"])
        for key, ok in (("true_skip", False), ("pad_mode", "reflect"),
                        ("activation", "ELU"), ("final_activation", None), ("trim_right_ratio", 1.0)):
            if conf.get(key, ok) != ok:
                raise _unsupported(f"{which}.{key}", conf[key])
    # every key of the 2-D nets is either plumbed into the architecture or refused here: nothing is silently ignored
    known_enc = {"ratios", "norm", "norm_params", "causal", "n_filters", "dimension", "n_residual_layers", "activation", "activation_params",
                 "kernel_size", "last_kernel_size", "residual_kernel_size", "dilation_base", "pad_mode", "true_skip", "compress", "seq_model",
                 "seq_layer_num", "res_seq", "conv_group_ratio", "input_size"}
    known_dec = (known_enc - {"dimension"}) | {"channels", "final_activation", "final_activation_params", "trim_right_ratio",
                                              "last_out_padding", "tr_conv_group_ratio"}
    for which, conf, known in (("encoder_conf", enc, known_enc), ("decoder_conf", dec, known_dec)):
        for key in conf:
            if key not in known:
                raise _unsupported(f"{which}.{key}", conf[key], "unknown key of the 2-D SEANet nets")
    lop = dec.get("last_out_padding", [(0, 1), (0, 0)])
    if [list(p_) for p_ in lop] != [[0, 1], [0, 0]]:
        raise _unsupported("decoder_conf.last_out_padding", lop, "the engine builds the default [(0, 1), (0, 0)] (seanet_decoder.py:260)")
    if dec.get("final_activation_params", None) not in (None, {}):
        raise _unsupported("decoder_conf.final_activation_params", dec.get("final_activation_params"))
    domain = list(m.get("codec_domain", ["time", "time"]))
    if domain not in (["mag_phase", "mag_phase"], ["mag_angle", "mag_angle"]):
        # `stft`: the reference's own decode branch keeps a channel axis the [:, :, :T] trim cannot take (codec_freq.py:413-415), no recipe
        # uses it; `mag` has no decode branch at all; mixed pairs have no recipe either
        raise _unsupported("model_conf.codec_domain", m.get("codec_domain"), "[mag_phase, mag_phase] and [mag_angle, mag_angle] are built")
    # mag_angle (conf/freqcodec_mag_angle_16k_n32_600k_step.yaml): the encoder sees torch.angle of the STFT.  Where a bin's imaginary part
    # is rounding noise around a negative real part (the first STFT frame, whose reflect padding makes it symmetric; DC / Nyquist rows) the
    # angle is +pi or -pi by the FFT's rounding, so the ENCODER INPUT of this domain is reproducible only modulo 2 pi -- by any second STFT,
    # the reference's own included (tests/golden/*ang*_variants: its fp64 STFT moves its own encoder output by the fixture's
    # `stft_self_noise`).  The engine runs the recipe; what its parity tests pin is said in tests/test_gpu_parity.py (features modulo 2 pi,
    # the whole path from the reference's features, the decode path from the reference's codes).
    n_in = 3 if domain[0] == "mag_phase" else 2
    if cfg.get("input_size", 1) != n_in or dec.get("channels", 1) != n_in:
        raise _unsupported("input_size/channels", (cfg.get("input_size", 1), dec.get("channels", 1)),
                           "mag_phase: 3 (log-magnitude, phase re, phase im); mag_angle: 2 (log-magnitude, angle)")

    def shared(key, default):
        a, b = enc.get(key, default), dec.get(key, default)
        if key == "ratios":
            a, b = [list(r) for r in a], [list(r) for r in b]
        if a != b:
            raise _unsupported(f"encoder_conf.{key} != decoder_conf.{key}", (a, b))
        return a

    ratios2 = shared("ratios", [[4, 1], [4, 1], [4, 2], [4, 1]])
    dimension = int(enc.get("dimension", 128))
    q0_ds_ratio = _q0_ds_ratio(q)
    # CostumeQuantizer's projection / tanh range (costume_quantizer.py:23-35), as for the time-domain codec
    codec_dim = q.get("codec_dim", None)
    codec_dim = dimension if codec_dim is None else int(codec_dim)
    if codec_dim not in (16, 32, 64, 128, 256, 512):
        raise _unsupported("quantizer_conf.codec_dim", q.get("codec_dim"), "the quantiser kernels are built for 16/32/64/128/256/512 dims")
    codec_range = q.get("codec_range", None)
    if codec_range is not None and not float(codec_range) > 0:
        raise _unsupported("quantizer_conf.codec_range", codec_range)
    if int(shared("n_residual_layers", 1)) != 1 and int(shared("dilation_base", 2)) != 1:
        raise _unsupported("n_residual_layers/dilation_base", (enc.get("n_residual_layers"), enc.get("dilation_base")))
    dc = dict(m.get("domain_conf", {}) or {})
    seg = m["segment_dur"] if "segment_dur" in m else 1.0                     # FreqCodec.__init__ defaults (codec_freq.py:142-143)
    ov = m["overlap_ratio"] if "overlap_ratio" in m else 0.01
    act_params = dict(shared("activation_params", {"alpha": 1.0}) or {})
    norm_params = dict(shared("norm_params", {}) or {})
    seq_model = shared("seq_model", "lstm")
    if seq_model not in ("lstm", "none", None):
        raise _unsupported("encoder_conf.seq_model", seq_model, "lstm or none")
    seq_model = "lstm" if seq_model == "lstm" else "none"
    arch = ArchSpec(
        sample_rate=int(m.get("target_sample_hz", 24000)), input_channels=n_in,
        audio_normalize=bool(m.get("audio_normalize", False)),            # FreqCodec.__init__ default is False (codec_freq.py:141)
        n_filters=int(shared("n_filters", 32)), dimension=dimension,
        ratios=tuple(int(r[1]) for r in ratios2), ratios_f=tuple(int(r[0]) for r in ratios2),
        kernel_size=int(shared("kernel_size", 7)), last_kernel_size=int(shared("last_kernel_size", 7)),
        residual_kernel_size=int(shared("residual_kernel_size", 3)), n_residual_layers=int(shared("n_residual_layers", 1)),
        dilation_base=int(shared("dilation_base", 2)), compress=int(shared("compress", 2)),
        lstm_layers=int(shared("seq_layer_num", 2)) if seq_model == "lstm" else 0, lstm_skip=bool(shared("res_seq", True)),
        elu_alpha=float(act_params.get("alpha", 1.0)), gn_eps=float(norm_params.get("eps", 1e-5)),
        codebook_size=int(q.get("codebook_size", 1024)), codebook_dim=codec_dim, num_quantizers=int(q.get("num_quantizers", 8)),
        codec_range=None if codec_range is None else float(codec_range),
        encoder_hop_length=int(q.get("encoder_hop_length", 320)), quantizer_sampling_rate=int(q.get("sampling_rate", 24000)),
        use_ddp=bool(q.get("use_ddp", True)), q0_ds_ratio=q0_ds_ratio, bypass_quantizer=bool(m.get("bypass_quantizer", False)),
        norm=str(shared("norm", "weight_norm")), causal=bool(shared("causal", False)),
        segment_dur=None if seg is None else float(seg), overlap_ratio=0.01 if ov is None else float(ov),
        model_type="freq_codec", n_fft=int(dc.get("n_fft", 512)), stft_hop=int(dc.get("hop_length", 160)),
        enc_conv_group_ratio=int(enc.get("conv_group_ratio", -1)), dec_conv_group_ratio=int(dec.get("conv_group_ratio", -1)),
        dec_tr_conv_group_ratio=int(dec.get("tr_conv_group_ratio", -1)),
    )
    if arch.stft_hop < 1 or arch.stft_hop > arch.n_fft // 2:
        # torch.istft checks the window envelope (NOLA, > 1e-11); a periodic Hann window's squared overlap-add reaches ~0 between
        # frames once the hop exceeds n_fft / 2, where the reference raises instead of returning audio
        raise _unsupported("model_conf.domain_conf.hop_length", arch.stft_hop, "must be in [1, n_fft / 2] (torch.istft's window-envelope check)")
    if arch.segment_dur is not None and not (arch.segment_dur > 0 and 0 <= arch.overlap_ratio < 1):
        raise _unsupported("model_conf.segment_dur/overlap_ratio", (arch.segment_dur, arch.overlap_ratio))
    if arch.segment_length is not None and arch.segment_length <= arch.n_fft // 2:
        raise _unsupported("model_conf.segment_dur", arch.segment_dur, "segments must be longer than n_fft / 2 samples (torch.stft reflect padding)")
    from .plan import encoder_plan_2d, decoder_plan_2d
    for op in encoder_plan_2d(arch) + decoder_plan_2d(arch):      # torch.nn.Conv2d's own constraints on `groups`
        if op.groups < 1 or op.cin % op.groups or op.cout % op.groups:
            raise _unsupported("conv_group_ratio", (arch.enc_conv_group_ratio, arch.dec_conv_group_ratio, arch.dec_tr_conv_group_ratio),
                               f"{op.key}: {op.groups} groups do not divide {op.cin} -> {op.cout} channels")
    f = arch.n_fft // 2 + 1
    for fr in reversed(arch.ratios_f):
        f = (f + fr - 2 * fr) // fr + 1          # SConv2d, kernel 2 fr, stride fr: padding_total = fr, no extra padding in frequency
    if f != 1:
        raise _unsupported("encoder_conf.ratios", ratios2, "the frequency ratios must reduce n_fft/2+1 bins to one (ReshapeModule)")
    return arch


def arch_from_config(cfg: Dict[str, Any]) -> ArchSpec:
    """Build an :class:`ArchSpec` from a dict loaded from the reference's ``config.yaml``."""
    if cfg.get("model", "encodec") == "freq_codec":
        return _freq_arch_from_config(cfg)
    if cfg.get("model", "encodec") != "encodec":
        raise _unsupported("model", cfg.get("model"))
    if cfg.get("encoder", "encodec_seanet_encoder") != "encodec_seanet_encoder":
        raise _unsupported("encoder", cfg.get("encoder"))
    if cfg.get("decoder", "encodec_seanet_decoder") != "encodec_seanet_decoder":
        raise _unsupported("decoder", cfg.get("decoder"))
    if cfg.get("quantizer", "costume_quantizer") != "costume_quantizer":
        raise _unsupported("quantizer", cfg.get("quantizer"))
    enc = _check_seanet_conf(cfg.get("encoder_conf", {}), "encoder_conf")
    dec = _check_seanet_conf(cfg.get("decoder_conf", {}), "decoder_conf")
    q = dict(cfg.get("quantizer_conf", {}) or {})
    m = dict(cfg.get("model_conf", {}) or {})

    def shared(key, default):
        a, b = enc.get(key, default), dec.get(key, default)
        if key == "ratios":
            a, b = list(a), list(b)
        if a != b:
            raise _unsupported(f"encoder_conf.{key} != decoder_conf.{key}", (a, b))
        return a

    # audio channels: the encoder's first conv takes `input_size` channels (gan_speech_codec.py:318-320, seanet_encoder.py:99), the decoder's
    # last conv emits decoder_conf.channels; Encodec._encode asserts <= 2 (codec_basic.py:344).  The two must agree for an encode -> decode model.
    input_size = cfg.get("input_size", 1)
    if input_size not in (1, 2) or dec.get("channels", 1) != input_size:
        raise _unsupported("input_size/channels", (input_size, dec.get("channels", 1)), "1 / 1 (mono) or 2 / 2 (stereo)")
    if m.get("codec_domain", "time") not in ("time", None):
        raise _unsupported("model_conf.codec_domain", m["codec_domain"])
    # the decoder's input_size is injected by build_model from quantizer.output_size()
    # (gan_speech_codec.py:325-329) == encoder dimension when codec_dim is unset
    dimension = int(enc.get("dimension", 128))
    codec_dim = q.get("codec_dim", None)
    codec_dim = dimension if codec_dim is None else int(codec_dim)
    if codec_dim not in (16, 32, 64, 128, 256, 512):
        raise _unsupported("quantizer_conf.codec_dim", q.get("codec_dim"), "the quantiser kernels are built for 16/32/64/128/256/512 dims")
    codec_range = q.get("codec_range", None)
    if codec_range is not None and not float(codec_range) > 0:
        raise _unsupported("quantizer_conf.codec_range", codec_range)
    q0_ds_ratio = _q0_ds_ratio(q)
    # decoder_conf values that differ from encoder_conf are refused (shared()), never silently ignored
    act_params = dict(shared("activation_params", {"alpha": 1.0}) or {})
    norm_params = dict(shared("norm_params", {}) or {})
    seq_model = shared("seq_model", "lstm")
    # Encodec.__init__ defaults for keys an ESPnet-style config.yaml may omit (codec_basic.py:132-141):
    # target_sample_hz=24000, audio_normalize=True, segment_dur=1.0, overlap_ratio=0.01.  An explicit null is kept.
    segment_dur = m["segment_dur"] if "segment_dur" in m else 1.0
    overlap_ratio = m["overlap_ratio"] if "overlap_ratio" in m else 0.01
    ratios = tuple(int(r) for r in shared("ratios", [8, 5, 4, 2]))
    arch = ArchSpec(
        sample_rate=int(m.get("target_sample_hz", 24000)),
        input_channels=int(input_size),
        audio_normalize=bool(m.get("audio_normalize", True)),
        n_filters=int(shared("n_filters", 32)),
        dimension=int(enc.get("dimension", 128)),
        ratios=ratios,
        kernel_size=int(shared("kernel_size", 7)),
        last_kernel_size=int(shared("last_kernel_size", 7)),
        residual_kernel_size=int(shared("residual_kernel_size", 3)),
        n_residual_layers=int(shared("n_residual_layers", 1)),
        dilation_base=int(shared("dilation_base", 2)),
        compress=int(shared("compress", 2)),
        lstm_layers=int(shared("seq_layer_num", 2)) if seq_model == "lstm" else 0,
        lstm_skip=bool(shared("res_seq", True)),
        elu_alpha=float(act_params.get("alpha", 1.0)),
        gn_eps=float(norm_params.get("eps", 1e-5)),
        codebook_size=int(q.get("codebook_size", 1024)),
        codebook_dim=codec_dim,
        codec_range=None if codec_range is None else float(codec_range),
        num_quantizers=int(q.get("num_quantizers", 8)),
        encoder_hop_length=int(q.get("encoder_hop_length", 320)),
        quantizer_sampling_rate=int(q.get("sampling_rate", 24000)),
        use_ddp=bool(q.get("use_ddp", True)),
        q0_ds_ratio=q0_ds_ratio,
        bypass_quantizer=bool(m.get("bypass_quantizer", False)),
        norm=str(shared("norm", "weight_norm")),
        causal=bool(shared("causal", False)),
        segment_dur=None if segment_dur is None else float(segment_dur),
        overlap_ratio=0.01 if overlap_ratio is None else float(overlap_ratio),
    )
    if arch.stft_hop < 1 or arch.stft_hop > arch.n_fft // 2:
        # torch.istft checks the window envelope (NOLA, > 1e-11); a periodic Hann window's squared overlap-add reaches ~0 between
        # frames once the hop exceeds n_fft / 2, where the reference raises instead of returning audio
        raise _unsupported("model_conf.domain_conf.hop_length", arch.stft_hop, "must be in [1, n_fft / 2] (torch.istft's window-envelope check)")
    if arch.segment_dur is not None and not (arch.segment_dur > 0 and 0 <= arch.overlap_ratio < 1):
        raise _unsupported("model_conf.segment_dur/overlap_ratio", (arch.segment_dur, arch.overlap_ratio))
    return arch


# ----------------------------------------------------------------------------------------------
# Recipe configs (the two architectures BASELINE.json names), as config.yaml-shaped dicts.
# Values follow egs/LibriTTS/codec/conf/encodec_16k_n32_600k_step_ds640.yaml:1-53 (ds640) and the
# class-default ratios of SEANetEncoder (seanet_encoder.py:92) for ds320.
# ----------------------------------------------------------------------------------------------
def fuzz_freq_recipe_config(seed: int) -> Dict[str, Any]:
    """A small pseudo-random FreqCodec (mag_phase) architecture: n_fft 64 / 128 / 512 with frequency ratios that reduce n_fft / 2 + 1 bins to
    one, time ratios 1 / 2, odd STFT hops included never (the engine wants an even hop), grouped and dense convs, with / without LSTM."""
    import random
    r = random.Random(7000 + seed)
    n_fft = r.choice([64, 128, 512])
    rf = {64: [4, 8], 128: [4, 4, 4], 512: [4, 4, 4, 4]}[n_fft]
    if n_fft == 64 and r.random() < 0.5:
        rf = [8, 4]
    ratios = [[f, r.choice([1, 1, 2])] for f in rf]
    nf = r.choice([4, 8])
    lstm = (nf << len(rf)) % 16 == 0 and r.random() < 0.6
    gr = r.choice([-1, -1, 1, 2]) if nf == 8 else -1
    nres = r.choice([1, 1, 2])
    dim = r.choice([16, 32])
    enc = {"ratios": ratios, "norm": "time_group_norm", "norm_params": {"num_groups": 1}, "causal": False, "dilation_base": 1,
           "n_filters": nf, "dimension": dim, "kernel_size": r.choice([3, 5, 7]), "last_kernel_size": r.choice([3, 5, 7]),
           "residual_kernel_size": 3, "compress": r.choice([1, 2]), "n_residual_layers": nres,
           "seq_model": "lstm" if lstm else "none", "seq_layer_num": r.choice([1, 2])}
    if gr > 0:
        enc["conv_group_ratio"] = gr
    if seed >= 1000:      # seeds from 1000 on (round 3; lower seeds keep their architectures: goldens are pinned on them) also draw the
        r2 = random.Random(9000 + seed)                       # norm / causality of the nets and the quantiser's projection
        if r2.random() < 0.6:
            enc["norm"] = "weight_norm"
            enc["causal"] = r2.random() < 0.5
            del enc["norm_params"]
    dec = dict({k: v for k, v in enc.items() if k != "dimension"}, channels=3)
    if gr > 0:
        dec["tr_conv_group_ratio"] = gr
    hop = r.choice([h for h in (n_fft // 4, n_fft // 2, 40, 160) if h <= n_fft // 2 and h % 2 == 0])
    tot = hop
    for _, t in ratios:
        tot *= t
    return {
        "input_size": 3, "sampling_rate": 16000,
        "encoder": "encodec_seanet_encoder_2d", "encoder_conf": enc,
        "quantizer": "costume_quantizer",
        "quantizer_conf": {"codebook_size": r.choice([64, 128]), "num_quantizers": r.choice([2, 4]), "ema_decay": 0.99, "kmeans_init": True,
                           "sampling_rate": 16000, "use_ddp": True, "encoder_hop_length": tot},
        "decoder": "encodec_seanet_decoder_2d", "decoder_conf": dec,
        "discriminator": "multiple_disc", "discriminator_conf": {"disc_conf_list": []},
        "model": "freq_codec",
        "model_conf": {"odim": dim, "multi_spectral_window_powers_of_two": [], "target_sample_hz": 16000,
                       "audio_normalize": r.random() < 0.6, "use_power_spec_loss": True, "segment_dur": None, "overlap_ratio": None,
                       "codec_domain": ["mag_phase", "mag_phase"], "domain_conf": {"n_fft": n_fft, "hop_length": hop}},
    }


def freq_recipe_config(name: str) -> Dict[str, Any]:
    """`freqmp`: egs/LibriTTS/codec/conf/freqcodec_mag_phase_16k_n32_600k_step.yaml:1-59 (16.2 M parameters);
    `freqmp640`: ..._ds640.yaml (time ratios 2,1,2,1, 640 samples per frame);
    `tinyfreq` / `tinyfreq640`: the same shapes with 4 base filters, 16-dim / 64-entry codebooks (small fixtures)."""
    if name.startswith("freqfuzz"):
        return fuzz_freq_recipe_config(int(name[8:]))
    q0 = name.endswith("q0")                          # quantizer_conf.q0_ds_ratio = 2: first stage on the half-rate sequence
    name = name[:-2] if q0 else name
    cd = name.endswith("cd")                          # CostumeQuantizer projection to codec_dim = 32 + tanh range (costume_quantizer.py:23-35)
    name = name[:-2] if cd else name
    wnc = name.endswith("wnc")                        # weight_norm + causal 2-D nets (conv.py:317-447 with causal = True)
    name = name[:-3] if wnc else name
    wn = name.endswith("wn")                          # weight_norm (no GroupNorm) 2-D nets
    name = name[:-2] if wn else name
    seg = name.endswith("seg")                        # FreqCodec._encode / _decode in segmented mode: 0.15 s frames, 10 % overlap
    name = name[:-3] if seg else name
    angle = name.endswith("ang")                      # freqcodec_mag_angle_16k_n32_600k_step.yaml: codec_domain [mag_angle, mag_angle], 2 channels
    name = name[:-3] if angle else name
    rel = name.endswith("rel")                        # "freqmpgr1rel": the one net of a hyper-parameter search that reproduces the README's
    name = name[:-3] if rel else name                 # 0.52 M parameters for the released gr1 model: n_filters 8, ONE LSTM layer
    gr = -1
    if "gr" in name:                                  # e.g. "freqmpgr1", "tinyfreqgr2": conv_group_ratio = tr_conv_group_ratio = N
        name, grs = name.split("gr")
        gr = int(grs)
    tiny = name.startswith("tinyfreq")
    if name not in ("freqmp", "tinyfreq", "freqmp640", "tinyfreq640"):
        raise KeyError(name)
    ds640 = name.endswith("640")
    ratios = [[4, 2], [4, 1], [4, 2], [4, 1]] if ds640 else [[4, 1], [4, 1], [4, 2], [4, 1]]
    enc = {"ratios": ratios, "norm": "time_group_norm", "norm_params": {"num_groups": 1}, "causal": False, "dilation_base": 1}
    if wn or wnc:
        enc = {"ratios": ratios, "norm": "weight_norm", "causal": bool(wnc), "dilation_base": 1}
    dec = dict(enc, channels=2 if angle else 3)
    if tiny:
        enc.update(n_filters=4, dimension=16)
        dec.update(n_filters=4)
    if gr > 0:
        if tiny:                                      # 8 base filters: every grouped layer keeps >= 2 groups
            enc.update(n_filters=8)
            dec.update(n_filters=8)
        enc.update(conv_group_ratio=gr)
        dec.update(conv_group_ratio=gr, tr_conv_group_ratio=gr)
    if rel:
        enc.update(n_filters=8, seq_model="lstm", seq_layer_num=1)
        dec.update(n_filters=8, seq_model="lstm", seq_layer_num=1)
    qc = {"codebook_size": 64 if tiny else 1024, "num_quantizers": 4 if tiny else 32, "ema_decay": 0.99,
          "kmeans_init": True, "sampling_rate": 16000, "quantize_dropout": True,
          "rand_num_quant": [1, 2, 4], "use_ddp": True, "encoder_hop_length": 640 if ds640 else 320}
    if cd:
        qc.update(codec_dim=32, codec_range=2.0)
    if q0:
        qc.update(q0_ds_ratio=2)
    return {
        "input_size": 2 if angle else 3, "sampling_rate": 16000,
        "encoder": "encodec_seanet_encoder_2d", "encoder_conf": enc,
        "quantizer": "costume_quantizer",
        "quantizer_conf": qc,
        "decoder": "encodec_seanet_decoder_2d", "decoder_conf": dec,
        "discriminator": "multiple_disc", "discriminator_conf": {"disc_conf_list": []},
        "model": "freq_codec",
        "model_conf": {"odim": 16 if tiny else 128, "multi_spectral_window_powers_of_two": [], "target_sample_hz": 16000,
                       "audio_normalize": True, "use_power_spec_loss": True, "segment_dur": 0.15 if seg else None,
                       "overlap_ratio": 0.1 if seg else None, "codec_domain": ["mag_angle", "mag_angle"] if angle else ["mag_phase", "mag_phase"]},
    }


def fuzz_recipe_config(seed: int) -> Dict[str, Any]:
    """A small pseudo-random architecture inside the engine's documented limits (the parity tests walk a few of these through the
    real reference / the oracle and the engine: unusual ratios, kernel sizes, widths, residual stacks, codebook sizes)."""
    import random
    r = random.Random(1000 + seed)
    ratios = [r.choice([2, 3, 4, 5, 8]) for _ in range(r.choice([1, 2, 3, 3, 4]))]
    lstm = r.random() < 0.6
    nf = r.choice([4, 8, 12]) if not lstm else r.choice([c for c in (4, 8, 12, 16) if (c << len(ratios)) % 16 == 0])
    norm = r.choice(["time_group_norm", "time_group_norm", "weight_norm"])
    causal = norm == "weight_norm" and r.random() < 0.5
    compress = r.choice([c for c in (1, 2, 4) if nf % c == 0])
    dim = r.choice([16, 32, 64])
    enc = {"ratios": ratios, "norm": norm, "causal": causal, "n_filters": nf, "dimension": dim,
           "kernel_size": r.choice([3, 5, 7]), "last_kernel_size": r.choice([3, 5, 7]), "residual_kernel_size": r.choice([3, 5]),
           "compress": compress, "n_residual_layers": r.choice([1, 1, 2, 3]), "dilation_base": r.choice([1, 2, 3]),
           "seq_model": "lstm" if lstm else "none", "seq_layer_num": r.choice([1, 2]),
           "activation_params": {"alpha": r.choice([1.0, 0.7])}}
    if norm == "time_group_norm":
        enc["norm_params"] = {"eps": r.choice([1e-5, 1e-3])}
    dec = {k: v for k, v in enc.items() if k != "dimension"}
    hop = 1
    for x in ratios:
        hop *= x
    cfg = {
        "input_size": 1, "sampling_rate": 16000,
        "encoder": "encodec_seanet_encoder", "encoder_conf": enc,
        "quantizer": "costume_quantizer",
        "quantizer_conf": {"codebook_size": r.choice([64, 128, 256]), "num_quantizers": r.choice([1, 3, 5, 8]), "ema_decay": 0.99,
                           "kmeans_init": True, "sampling_rate": 16000, "use_ddp": True, "encoder_hop_length": hop},
        "decoder": "encodec_seanet_decoder", "decoder_conf": dec,
        "discriminator": "multiple_disc", "discriminator_conf": {"disc_conf_list": []},
        "model": "encodec",
        "model_conf": {"odim": dim, "multi_spectral_window_powers_of_two": [], "target_sample_hz": 16000,
                       "audio_normalize": r.random() < 0.7, "use_power_spec_loss": True, "segment_dur": None, "overlap_ratio": None},
    }
    if seed >= 3000:          # round 4 (drawn last: the committed seeds keep their architectures): stereo models, first stage at half rate
        if r.random() < 0.5:
            cfg["input_size"] = 2
            dec["channels"] = 2
        if r.random() < 0.5:
            cfg["quantizer_conf"]["q0_ds_ratio"] = r.choice([2, 2, 4])
    return cfg


def recipe_config(name: str) -> Dict[str, Any]:
    if name.startswith("fuzz"):
        return fuzz_recipe_config(int(name[4:]))
    if name.startswith(("freqmp", "tinyfreq", "freqfuzz")):
        return freq_recipe_config(name)
    if name in ("tinyst", "ds320st", "tinystwn", "ds320stseg"):
        # stereo (the reference's "48 kHz" flavour, seanet_encoder.py:63-65: channels = 2): input_size 2, decoder_conf.channels 2; volume scale
        # from the channel mean (codec_basic.py:366-371).  "tinystwn": weight_norm causal convs; "ds320stseg": segmented overlap-add
        cfg = recipe_config({"tinyst": "tiny", "ds320st": "ds320", "tinystwn": "tinywn", "ds320stseg": "ds320seg"}[name])
        cfg["input_size"] = 2
        cfg["decoder_conf"]["channels"] = 2
        return cfg
    if name in ("tinybypass", "tinybypassseg"):          # model_conf.bypass_quantizer (codec_basic.py:700-701), also in segmented mode
        cfg = recipe_config("tiny")
        cfg["model_conf"]["bypass_quantizer"] = True
        if name.endswith("seg"):
            cfg["model_conf"]["segment_dur"] = 0.05          # 800-sample frames
            cfg["model_conf"]["overlap_ratio"] = 0.1
        return cfg
    if name in ("tinyq0", "ds320q0", "ss320q0"):
        # quantizer_conf.q0_ds_ratio > 1 (ddp_core_vq.py:354-356,396-404): first stage on the half-rate sequence.  "tinyq0" asks for 3:
        # the reference halves whatever the value is, and so must the engine
        cfg = recipe_config({"tinyq0": "tiny", "ds320q0": "ds320", "ss320q0": "ss320"}[name])
        cfg["quantizer_conf"]["q0_ds_ratio"] = 3 if name == "tinyq0" else 2
        return cfg
    if name in ("ds320cd64", "tinycd"):   # CostumeQuantizer with codec_dim != input_size and a tanh range (costume_quantizer.py:23-35)
        cfg = recipe_config("ds320" if name == "ds320cd64" else "tiny")
        cfg["quantizer_conf"]["codec_dim"] = 64 if name == "ds320cd64" else 32
        cfg["quantizer_conf"]["codec_range"] = 2.5
        return cfg
    if name == "tinyrange":               # tanh range without a projection
        cfg = recipe_config("tiny")
        cfg["quantizer_conf"]["codec_range"] = 1.5
        return cfg
    if name == "ds320seg":   # ds320 run in the segmented overlap-add mode (0.5 s frames, 10 % overlap)
        cfg = recipe_config("ds320")
        cfg["model_conf"]["segment_dur"] = 0.5
        cfg["model_conf"]["overlap_ratio"] = 0.1
        return cfg
    if name == "ds640seg":   # segment length NOT a multiple of the hop (8000 % 640 = 320): frames decode to 13 * 640 = 8320 samples
        cfg = recipe_config("ds640")
        cfg["model_conf"]["segment_dur"] = 0.5
        cfg["model_conf"]["overlap_ratio"] = 0.1
        return cfg
    if name in ("ss320nc", "tinyssnc", "ss640nc"):
        # egs/LibriTTS/codec/conf/soundstream_noncausal_16k_n32_600k_step.yaml:11-38: GroupNorm, non-causal, three residual
        # blocks per stage (dilations 1, 2, 4 -> the two-source GroupNorm chain across consecutive blocks), no sequence
        # model, 512-dim codebooks ("tinyssnc": the same shape, small)
        cfg = recipe_config({"ss320nc": "ds320", "ss640nc": "ds640"}.get(name, "tiny"))      # ss640nc: ..._step_ds640.yaml
        for k in ("encoder_conf", "decoder_conf"):
            cfg[k]["n_residual_layers"] = 3
            cfg[k]["seq_model"] = "none"
        cfg["encoder_conf"]["dimension"] = 32 if name == "tinyssnc" else 512
        cfg["model_conf"]["odim"] = cfg["encoder_conf"]["dimension"]
        return cfg
    if name in ("ss320", "tinyss"):
        # egs/LibriTTS/codec/conf/soundstream_16k_n32_600k_step.yaml:11-38: weight-normalised causal convs, three residual
        # blocks per stage with dilations 1, 2, 4, no sequence model, 512-dim codebooks ("tinyss": the same shape, small)
        cfg = recipe_config("ds320wn" if name == "ss320" else "tinywn")
        for k in ("encoder_conf", "decoder_conf"):
            cfg[k]["n_residual_layers"] = 3
            cfg[k]["seq_model"] = "none"
        cfg["encoder_conf"]["dimension"] = 512 if name == "ss320" else 32
        return cfg
    if name in ("ds320wn", "tinywn"):   # weight-normalised, causal variants of the same nets (EnCodec-style streaming convs)
        cfg = recipe_config("ds320" if name == "ds320wn" else "tiny")
        for k in ("encoder_conf", "decoder_conf"):
            cfg[k]["norm"] = "weight_norm"
            cfg[k]["causal"] = True
            cfg[k].pop("norm_params", None)
        return cfg
    if name == "ds640":
        ratios, hop = [8, 5, 4, 2, 2], 640
    elif name == "ds320":
        ratios, hop = [8, 5, 4, 2], 320
    elif name == "tiny":  # small architecture used by the committed golden fixtures
        return {
            "input_size": 1, "sampling_rate": 16000,
            "encoder": "encodec_seanet_encoder",
            "encoder_conf": {"ratios": [4, 2], "norm": "time_group_norm", "causal": False,
                             "n_filters": 8, "dimension": 16},
            "quantizer": "costume_quantizer",
            "quantizer_conf": {"codebook_size": 64, "num_quantizers": 6, "ema_decay": 0.99,
                               "kmeans_init": True, "sampling_rate": 16000, "use_ddp": True,
                               "encoder_hop_length": 8},
            "decoder": "encodec_seanet_decoder",
            "decoder_conf": {"ratios": [4, 2], "norm": "time_group_norm", "causal": False,
                             "n_filters": 8},
            "discriminator": "multiple_disc",
            "discriminator_conf": {"disc_conf_list": []},
            "model": "encodec",
            "model_conf": {"odim": 16, "multi_spectral_window_powers_of_two": [],
                           "target_sample_hz": 16000, "audio_normalize": True,
                           "use_power_spec_loss": True, "segment_dur": None, "overlap_ratio": None},
        }
    else:
        raise KeyError(name)
    return {
        "input_size": 1, "sampling_rate": 16000,
        "encoder": "encodec_seanet_encoder",
        "encoder_conf": {"ratios": list(ratios), "norm": "time_group_norm", "causal": False},
        "quantizer": "costume_quantizer",
        "quantizer_conf": {"codebook_size": 1024, "num_quantizers": 32, "ema_decay": 0.99,
                           "kmeans_init": True, "sampling_rate": 16000, "quantize_dropout": True,
                           "rand_num_quant": [2, 4, 8, 16, 32], "use_ddp": True,
                           "encoder_hop_length": hop},
        "decoder": "encodec_seanet_decoder",
        "decoder_conf": {"ratios": list(ratios), "norm": "time_group_norm", "causal": False},
        "discriminator": "multiple_disc",
        "discriminator_conf": {"disc_conf_list": [
            {"name": "encodec_multi_scale_stft_discriminator", "filters": 32}]},
        "model": "encodec",
        "model_conf": {"odim": 128, "multi_spectral_window_powers_of_two": [],
                       "target_sample_hz": 16000, "audio_normalize": True,
                       "use_power_spec_loss": True, "segment_dur": None, "overlap_ratio": None},
    }
