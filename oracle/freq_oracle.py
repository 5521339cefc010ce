"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product path `funcodec_amd/`.

CPU restatement of the reference's FreqCodec path (BASELINE.json configs[3]): STFT -> log-magnitude /
phase image -> 2-D SEANet encoder -> residual vector quantiser -> 2-D SEANet decoder -> softplus(magnitude) * phase -> inverse
STFT, written as straight-line functional code over the same ATen CPU operators the reference calls.  Pinned bit-for-bit against
the real reference by `oracle/make_golden.py` (cases of kind "freq"); the engine does NOT implement this path yet
(funcodec_amd/config.py refuses `model: freq_codec`) -- this file and its fixtures are the first step (oracle before kernels).

Reference lines restated (paths relative to /root/reference):
  FreqCodec._encode_frame / _decode_frame      funcodec/models/codec_freq.py:330-448
  torchaudio.transforms.Spectrogram / InverseSpectrogram: third-party, absent from this image; restated from torchaudio's
      published implementation (functional.spectrogram / inverse_spectrogram are thin wrappers over torch.stft / torch.istft with
      center=True, pad_mode="reflect", a periodic Hann window of n_fft samples, normalized=False, onesided=True)
  SConv2d / pad2d / SConvTranspose2d / unpad2d funcodec/modules/normed_modules/conv.py:100-141,317-447
  SEANetResnetBlock2d / SEANetEncoder2d        funcodec/models/encoder/seanet_encoder.py:188-363
  SEANetDecoder2d                              funcodec/models/decoder/seanet_decoder.py:183-360
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from torch_oracle import Oracle, get_extra_padding_for_conv1d, sconv1d


def spectrogram(x: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """torchaudio.transforms.Spectrogram(n_fft, hop_length=hop, power=None): [B,T] -> complex [B, n_fft/2+1, 1 + T//hop]."""
    win = torch.hann_window(n_fft)
    return torch.stft(x, n_fft, hop, n_fft, win, center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)


def inverse_spectrogram(s: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """torchaudio.transforms.InverseSpectrogram(n_fft, hop_length=hop): complex [B,F,T'] -> [B, hop*(T'-1)]."""
    win = torch.hann_window(n_fft)
    return torch.istft(s, n_fft, hop, n_fft, win, center=True, normalized=False, onesided=True, length=None, return_complex=False)


def pad2d_reflect(x: torch.Tensor, pad_time: Tuple[int, int], pad_freq: Tuple[int, int]) -> torch.Tensor:
    """conv.py:100-119 (mode == 'reflect'): zero-extend short dims first, reflect, trim the extension."""
    freq_len, time_len = x.shape[-2:]
    max_t, max_f = max(pad_time), max(pad_freq)
    extra_t = max_t - time_len + 1 if time_len <= max_t else 0
    extra_f = max_f - freq_len + 1 if freq_len <= max_f else 0
    x = F.pad(x, [0, extra_t, 0, extra_f])
    padded = F.pad(x, (*pad_time, *pad_freq), "reflect")
    return padded[..., :padded.shape[-2] - extra_f, :padded.shape[-1] - extra_t]


def sconv2d(x, w, b, gamma, beta, stride: Tuple[int, int], eps: float, dilation: Tuple[int, int] = (1, 1), causal: bool = False):
    """SConv2d.forward conv.py:342-381 -> NormConv2d -> GroupNorm(1, C) over (C, F, T) (or none: weight_norm).  `causal` concerns the
    time axis only (:361-367): all of the fixed padding before, the extra padding after; the frequency axis is padded as always."""
    kf, kt = w.shape[-2:]
    tot_f = (kf - 1) * dilation[0] - (stride[0] - 1)
    tot_t = (kt - 1) * dilation[1] - (stride[1] - 1)
    extra_t = get_extra_padding_for_conv1d(x.shape[-1], kt, stride[1], tot_t)       # no extra padding on the frequency axis (:354-356)
    f_after = tot_f // 2
    f_before = tot_f - f_after
    t_after = tot_t // 2
    t_before = tot_t - t_after + extra_t          # NB: the reference adds the extra padding on the LEFT of the time axis here (:377)
    if causal:
        t_before, t_after = tot_t, extra_t
    x = pad2d_reflect(x, (t_before, t_after), (f_before, f_after))
    y = F.conv2d(x, w, b, stride=stride, dilation=dilation, groups=x.shape[1] // w.shape[1])     # conv_group_ratio > 0: grouped
    return y if gamma is None else F.group_norm(y, 1, gamma, beta, eps)


def sconvtr2d(x, w, b, gamma, beta, stride: Tuple[int, int], eps: float, out_padding=((0, 0), (0, 0)), causal: bool = False):
    """SConvTranspose2d.forward conv.py:408-447: ConvTranspose2d -> GroupNorm on the untrimmed output (or none) -> unpad2d,
    the trims reduced by `out_padding` ([(freq_left, freq_right), (time_left, time_right)]).  `causal` (trim_right_ratio = 1, :427-431):
    the whole time trim on the right."""
    kf, kt = w.shape[-2:]
    y = F.conv_transpose2d(x, w, b, stride=stride, groups=b.shape[0] // w.shape[1])             # tr_conv_group_ratio > 0
    if gamma is not None:
        y = F.group_norm(y, 1, gamma, beta, eps)
    pf, pt = kf - stride[0], kt - stride[1]
    f_r, t_r = pf // 2, pt // 2
    if causal:
        t_r = pt
    f_l, t_l = pf - f_r, pt - t_r
    (fo_l, fo_r), (to_l, to_r) = out_padding
    f_l, f_r, t_l, t_r = max(f_l - fo_l, 0), max(f_r - fo_r, 0), max(t_l - to_l, 0), max(t_r - to_r, 0)
    return y[..., f_l: y.shape[-2] - f_r, t_l: y.shape[-1] - t_r]


class FreqOracle(Oracle):
    """FreqCodec built from a config.yaml-shaped dict and a reference-format state_dict."""

    def __init__(self, config: Dict, state: Dict[str, torch.Tensor]):
        super().__init__(config, state)
        enc = dict(config.get("encoder_conf", {}))
        m = dict(config.get("model_conf", {}))
        self.ratios2d: List[Tuple[int, int]] = [tuple(r) for r in enc.get("ratios", [[4, 1], [4, 1], [4, 2], [4, 1]])]
        self.domain = tuple(m.get("codec_domain", ("time", "time")))
        assert self.domain in (("mag_phase", "mag_phase"), ("mag_angle", "mag_angle")), "the mag_phase and mag_angle recipes are restated"
        dc = dict(m.get("domain_conf", {}) or {})
        self.n_fft, self.stft_hop = dc.get("n_fft", 512), dc.get("hop_length", 160)
        self.audio_normalize = m.get("audio_normalize", False)          # FreqCodec.__init__ default is False (codec_freq.py:141)
        self.in_ch = config.get("input_size", 3)
        self.last_out_padding = ((0, 1), (0, 0))                       # SEANetDecoder2d default (seanet_decoder.py:279)

    def _conv2(self, x, prefix, stride=(1, 1), dilation=(1, 1)):
        w, b, g, be = self._p(prefix)
        return sconv2d(x, w, b, g, be, stride, self.eps, dilation, self.causal)

    def _resblock2(self, x, prefix, dil_t=1):
        """SEANetResnetBlock2d.forward seanet_encoder.py:239-240: shortcut(x) + block(x); block = ELU, 3x3 (dilation (1, d)), ELU, 1x1."""
        y = self._conv2(self._elu(x), f"{prefix}.block.1.conv", dilation=(1, dil_t))
        y = self._conv2(self._elu(y), f"{prefix}.block.3.conv")
        return self._conv2(x, f"{prefix}.shortcut.conv") + y

    @torch.no_grad()
    def encoder2d(self, x: torch.Tensor) -> torch.Tensor:
        """SEANetEncoder2d.forward seanet_encoder.py:359-363: [B,3,F,T'] -> [B,Tf,D]."""
        idx = 0
        x = self._conv2(x, f"encoder.model.{idx}.conv")
        idx += 1
        for fr, tr in reversed(self.ratios2d):
            for j in range(self.n_res):
                x = self._resblock2(x, f"encoder.model.{idx}", self.dil_base ** j)
                idx += 1
            idx += 1                                                   # ELU
            x = self._conv2(self._elu(x), f"encoder.model.{idx}.conv", stride=(fr, tr))
            idx += 1
        assert x.shape[2] == 1, "the frequency axis must be reduced to 1 bin (ReshapeModule squeezes it)"
        x = x.squeeze(2)
        idx += 1                                                       # ReshapeModule
        if self.lstm_layers > 0:
            x = self._slstm(x, f"encoder.model.{idx}.lstm")
            idx += 1
        idx += 1                                                       # ELU
        x = self._conv(self._elu(x), f"encoder.model.{idx}.conv")
        return x.permute(0, 2, 1)

    @torch.no_grad()
    def decoder2d(self, z: torch.Tensor) -> torch.Tensor:
        """SEANetDecoder2d.forward seanet_decoder.py:357-360: [B,Tf,D] -> [B,3,F,T']."""
        x = z.permute(0, 2, 1)
        idx = 0
        x = self._conv(x, f"decoder.model.{idx}.conv")
        idx += 1
        if self.lstm_layers > 0:
            x = self._slstm(x, f"decoder.model.{idx}.lstm")
            idx += 1
        x = x.unsqueeze(2)
        idx += 1                                                       # ReshapeModule
        n = len(self.ratios2d)
        for i, (fr, tr) in enumerate(self.ratios2d):
            idx += 1                                                   # ELU
            w, b, g, be = self._p(f"decoder.model.{idx}.convtr")
            x = sconvtr2d(self._elu(x), w, b, g, be, (fr, tr), self.eps, self.last_out_padding if i == n - 1 else ((0, 0), (0, 0)), self.causal)
            idx += 1
            for j in range(self.n_res):
                x = self._resblock2(x, f"decoder.model.{idx}", self.dil_base ** j)
                idx += 1
        idx += 1                                                       # ELU
        return self._conv2(self._elu(x), f"decoder.model.{idx}.conv")

    @torch.no_grad()
    def encode_frame(self, speech: torch.Tensor):
        """FreqCodec._encode_frame codec_freq.py:330-392 (mag_phase / mag_angle): speech [B,1,T] -> emb [B,Tf,D], scale [B,1]|None, features."""
        x = speech
        scale = None
        if self.audio_normalize:
            mono = x.mean(dim=1, keepdim=True)
            volume = mono.pow(2).mean(dim=2, keepdim=True).sqrt()
            scale = 1e-8 + volume
            x = x / scale
            scale = scale.view(-1, 1)
        xc = spectrogram(x.squeeze(1), self.n_fft, self.stft_hop)
        feats = self.features(xc)
        return self.encoder2d(feats), scale, feats

    def features(self, xc: torch.Tensor) -> torch.Tensor:
        """codec_freq.py:356-379: the 2-D encoder's input from the complex STFT (encoder.input_size 2 / 3: the stacked form)."""
        mag = torch.abs(xc)
        log_mag = torch.log(torch.clamp(mag, min=1e-6))
        if self.domain[0] == "mag_angle":                               # :356-364
            return torch.stack([log_mag, torch.angle(xc)], dim=1)
        phase = xc / torch.clamp(mag, min=1e-6)                         # :371-379
        return torch.stack([log_mag, phase.real, phase.imag], dim=1)

    @torch.no_grad()
    def decode_frame(self, emb: torch.Tensor, scale):
        """FreqCodec._decode_frame codec_freq.py:409-448 (mag_phase): softplus(mag) * (re + i im) -> inverse STFT -> x scale."""
        out = self.decoder2d(emb)
        parts = [p.squeeze(1) for p in torch.split(out, 1, dim=1)]
        mag = F.softplus(parts[0])
        if self.domain[1] == "mag_angle":                               # codec_freq.py:426-434
            ang = torch.sin(parts[1]) * torch.pi
            spec = torch.complex(torch.cos(ang) * mag, torch.sin(ang) * mag)
        else:
            spec = mag * torch.complex(parts[1], parts[2])
        wav = inverse_spectrogram(spec, self.n_fft, self.stft_hop).unsqueeze(1)
        if scale is not None:
            wav = wav * scale.view(-1, 1, 1)
        return wav, out

    @torch.no_grad()
    def inference(self, speech: torch.Tensor, bit_width=None, use_scale=True, need_recon=True):
        """FreqCodec.inference codec_freq.py (one frame: segment_dur null)."""
        if speech.dim() == 2:
            speech = speech.unsqueeze(1)
        m = self.cfg.get("model_conf", {})
        if (m["segment_dur"] if "segment_dur" in m else 1.0) is not None:      # FreqCodec._encode / _decode codec_freq.py:303-328,390-404
            seg = int(m.get("segment_dur", 1.0) * int(m.get("target_sample_hz", 24000)))
            ov = 0.01 if m.get("overlap_ratio", 0.01) is None else m.get("overlap_ratio", 0.01)
            stride = max(1, int((1 - ov) * seg))
            T = speech.shape[-1]
            idxs, embs, subs_all, recons, encs, scales = [], [], [], [], [], []
            for off in range(0, T, stride):
                emb, scale, _ = self.encode_frame(speech[:, :, off:off + seg])
                quant, idx, subs = self.rvq_forward(emb, self.n_q_for(bit_width))
                idxs.append(idx); embs.append((quant, scale if use_scale else None)); subs_all.append(subs)
                encs.append(emb); scales.append(scale)
                if need_recon:
                    recons.append(self.decode_frame(quant, scale if use_scale else None)[0])
            recon = self.linear_overlap_add(recons, stride)[:, :, :T] if need_recon else None
            return dict(code_indices=idxs, code_embeddings=embs, recon_speech=recon, sub_quants=subs_all, encoder_out=encs, scale=scales)
        emb, scale, feats = self.encode_frame(speech)
        quant, idx, subs = self.rvq_forward(emb, self.n_q_for(bit_width))
        recon, dec_out = None, None
        if need_recon:
            recon, dec_out = self.decode_frame(quant, scale if use_scale else None)
            recon = recon[:, :, :speech.shape[-1]]
        return dict(code_indices=[idx], code_embeddings=[(quant, scale if use_scale else None)], recon_speech=recon,
                    sub_quants=[subs], encoder_out=emb, scale=scale, features=feats, decoder_out=dec_out)
