"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product path `funcodec_amd/`.

CPU restatement of the reference's encode/decode hot path, written as straight-line functional
code over the *same ATen CPU operators the reference calls* (`F.conv1d`, `F.conv_transpose1d`,
`F.group_norm`, `F.elu`, `F.pad`, `torch._VF.lstm` via `nn.LSTM`, `@`, `max`, `F.embedding`), in the
same order, so that on the same torch build / thread count it is bit-identical to the reference
modules.  It needs nothing from /root/reference and therefore travels to the GPU box, where it is
the checker for the `-m gpu` parity tests and the timed `cpu_baseline` (kind "port") of bench.py.

Pinned against the real reference in this container by `oracle/make_golden.py` (bit-exact indices
and waveforms for every fixture, see tests/golden/MANIFEST.json) and by tests/test_oracle.py.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# padding arithmetic -- funcodec/modules/normed_modules/conv.py
# ------------------------------------------------------------------------------------------------
def get_extra_padding_for_conv1d(length: int, kernel_size: int, stride: int, padding_total: int) -> int:
    """conv.py:57-64"""
    n_frames = (length - kernel_size + padding_total) / stride + 1
    ideal_length = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
    return ideal_length - length


def pad1d_reflect(x: torch.Tensor, paddings: Tuple[int, int]) -> torch.Tensor:
    """conv.py:82-99 (mode == 'reflect'): zero-extend short inputs first, reflect, trim the extension."""
    length = x.shape[-1]
    pl, pr = paddings
    max_pad = max(pl, pr)
    extra_pad = 0
    if length <= max_pad:
        extra_pad = max_pad - length + 1
        x = F.pad(x, (0, extra_pad))
    padded = F.pad(x, (pl, pr), "reflect")
    end = padded.shape[-1] - extra_pad
    return padded[..., :end]


def sconv1d(x, w, b, gamma, beta, stride: int, eps: float, causal: bool = False, dilation: int = 1):
    """SConv1d.forward conv.py:243-261 -> NormConv1d.forward :155-164 -> GroupNorm(1,C) :45-52 (gamma None: no output
    norm, i.e. norm = weight_norm / none)."""
    k = w.shape[-1]
    padding_total = (k - 1) * dilation - (stride - 1)             # :247
    extra = get_extra_padding_for_conv1d(x.shape[-1], k, stride, padding_total)
    if causal:
        x = pad1d_reflect(x, (padding_total, extra))              # :249-251
    else:
        pr = padding_total // 2
        pl = padding_total - pr
        x = pad1d_reflect(x, (pl, pr + extra))
    y = F.conv1d(x, w, b, stride=stride, dilation=dilation)
    return y if gamma is None else F.group_norm(y, 1, gamma, beta, eps)


def sconvtr1d(x, w, b, gamma, beta, stride: int, eps: float, causal: bool = False):
    """SConvTranspose1d.forward conv.py:281-305: ConvTranspose1d -> GroupNorm on the UNTRIMMED output -> unpad1d
    (causal, trim_right_ratio = 1: everything trimmed on the right :292-297)."""
    k = w.shape[-1]
    y = F.conv_transpose1d(x, w, b, stride=stride)
    if gamma is not None:
        y = F.group_norm(y, 1, gamma, beta, eps)
    padding_total = k - stride
    if causal:
        pr = math.ceil(padding_total * 1.0)
        pl = padding_total - pr
    else:
        pr = padding_total // 2
        pl = padding_total - pr
    return y[..., pl: y.shape[-1] - pr]


def weight_norm_fold(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """torch.nn.utils.weight_norm (dim = 0): the weight the module computes in its forward pre-hook."""
    return torch._weight_norm(v, g, 0)


class Oracle:
    """Functional model built from a config.yaml-shaped dict and a reference-format state_dict."""

    def __init__(self, config: Dict, state: Dict[str, torch.Tensor]):
        self.cfg = config
        enc, dec = dict(config.get("encoder_conf", {})), dict(config.get("decoder_conf", {}))
        q, m = dict(config.get("quantizer_conf", {})), dict(config.get("model_conf", {}))
        self.norm = enc.get("norm", "weight_norm")
        self.causal = bool(enc.get("causal", False))
        assert self.norm in ("time_group_norm", "weight_norm", "none") and dec.get("norm", "weight_norm") == self.norm
        assert bool(dec.get("causal", False)) == self.causal
        self.ratios: List[int] = list(enc.get("ratios", [8, 5, 4, 2]))
        assert list(dec.get("ratios", [8, 5, 4, 2])) == self.ratios
        self.n_filters = enc.get("n_filters", 32)
        self.dimension = enc.get("dimension", 128)
        self.ksize = enc.get("kernel_size", 7)
        self.last_ksize = enc.get("last_kernel_size", 7)
        self.res_ksize = enc.get("residual_kernel_size", 3)
        self.compress = enc.get("compress", 2)
        self.n_res = enc.get("n_residual_layers", 1)
        self.dil_base = enc.get("dilation_base", 2)
        self.lstm_layers = enc.get("seq_layer_num", 2) if enc.get("seq_model", "lstm") == "lstm" else 0
        self.alpha = (enc.get("activation_params") or {"alpha": 1.0}).get("alpha", 1.0)
        self.eps = (enc.get("norm_params") or {}).get("eps", 1e-5)
        self.audio_normalize = m.get("audio_normalize", True)      # Encodec.__init__ default, codec_basic.py:139
        self.codebook_size = q.get("codebook_size", 1024)
        self.num_quantizers = q.get("num_quantizers", 8)
        self.hop = q.get("encoder_hop_length", 320)
        self.q_sr = q.get("sampling_rate", 24000)
        self.sd = {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(v)).float() for k, v in state.items()
                   if k.startswith(("encoder.", "decoder.", "quantizer."))}
        self.embed = self.sd["quantizer.rq.model.embed"]            # [n_q, K, D]
        self._lstms: Dict[str, torch.nn.LSTM] = {}

    # -- small helpers -------------------------------------------------------------------------
    def _p(self, prefix):
        inner = "convtr" if prefix.endswith("convtr") else "conv"
        if self.norm == "weight_norm":
            w = weight_norm_fold(self.sd[f"{prefix}.{inner}.weight_v"], self.sd[f"{prefix}.{inner}.weight_g"])
        else:
            w = self.sd[f"{prefix}.{inner}.weight"]
        if self.norm == "time_group_norm":
            return w, self.sd[f"{prefix}.{inner}.bias"], self.sd[f"{prefix}.norm.weight"], self.sd[f"{prefix}.norm.bias"]
        return w, self.sd[f"{prefix}.{inner}.bias"], None, None

    def _conv(self, x, prefix, stride=1, dilation=1):
        w, b, g, be = self._p(prefix)
        return sconv1d(x, w, b, g, be, stride, self.eps, self.causal, dilation)

    def _elu(self, x):
        return F.elu(x, self.alpha)                                  # activations.py:24-30

    def _resblock(self, x, prefix, dilation=1):
        """SEANetResnetBlock.forward seanet_encoder.py:60-61: shortcut(x) + block(x)."""
        y = self._conv(self._elu(x), f"{prefix}.block.1.conv", dilation=dilation)
        y = self._conv(self._elu(y), f"{prefix}.block.3.conv")
        return self._conv(x, f"{prefix}.shortcut.conv") + y

    def _slstm(self, x, prefix):
        """SLSTM.forward lstm.py:22-28."""
        if prefix not in self._lstms:
            h = x.shape[1]
            lstm = torch.nn.LSTM(h, h, self.lstm_layers)
            with torch.no_grad():
                for name, p in lstm.named_parameters():
                    p.copy_(self.sd[f"{prefix}.{name}"])
            self._lstms[prefix] = lstm.eval()
        xt = x.permute(2, 0, 1)
        y, _ = self._lstms[prefix](xt)
        y = y + xt
        return y.permute(1, 2, 0)

    # -- encoder / decoder ---------------------------------------------------------------------
    @torch.no_grad()
    def encoder(self, x: torch.Tensor) -> torch.Tensor:
        """SEANetEncoder.forward seanet_encoder.py:171-185: [B,1,T] -> [B,Tf,D]."""
        idx = 0
        x = self._conv(x, f"encoder.model.{idx}.conv")
        idx += 1
        for ratio in reversed(self.ratios):
            for j in range(self.n_res):
                x = self._resblock(x, f"encoder.model.{idx}", self.dil_base ** j)
                idx += 1
            idx += 1
            x = self._conv(self._elu(x), f"encoder.model.{idx}.conv", stride=ratio)
            idx += 1
        if self.lstm_layers > 0:
            x = self._slstm(x, f"encoder.model.{idx}.lstm")
            idx += 1
        idx += 1
        x = self._conv(self._elu(x), f"encoder.model.{idx}.conv")
        return x.permute(0, 2, 1)

    @torch.no_grad()
    def decoder(self, z: torch.Tensor) -> torch.Tensor:
        """SEANetDecoder.forward seanet_decoder.py:177-180: [B,Tf,D] -> [B,1,Tf*hop]."""
        x = z.permute(0, 2, 1)
        idx = 0
        x = self._conv(x, f"decoder.model.{idx}.conv")
        idx += 1
        if self.lstm_layers > 0:
            x = self._slstm(x, f"decoder.model.{idx}.lstm")
            idx += 1
        for ratio in self.ratios:
            idx += 1
            w, b, g, be = self._p(f"decoder.model.{idx}.convtr")
            x = sconvtr1d(self._elu(x), w, b, g, be, ratio, self.eps, self.causal)
            idx += 1
            for j in range(self.n_res):
                x = self._resblock(x, f"decoder.model.{idx}", self.dil_base ** j)
                idx += 1
        idx += 1
        return self._conv(self._elu(x), f"decoder.model.{idx}.conv")

    # -- quantizer -----------------------------------------------------------------------------
    def n_q_for(self, bandwidth) -> int:
        """vq.py:105-117."""
        bw_per_q = math.log2(self.codebook_size) * self.q_sr / self.hop
        n_q = self.num_quantizers
        if bandwidth and bandwidth > 0.0:
            n_q = int(max(1, math.floor(bandwidth / bw_per_q)))
        return min(n_q, self.num_quantizers)

    @torch.no_grad()
    def rvq_forward(self, emb: torch.Tensor, n_q: int):
        """DistributedResidualVectorQuantization.forward ddp_core_vq.py:367-418 (eval branch) through
        CostumeQuantizer.inference costume_quantizer.py:77-96.  emb [B,Tf,D] ->
        quantized [B,Tf,D], indices [n_q,B,Tf] i64, sub_quants [n_q,B,D,Tf]."""
        qc = self.cfg.get("quantizer_conf", {})
        if "quantizer.input_proj.weight" in self.sd:              # CostumeQuantizer.inference :84-87
            emb = F.linear(emb, self.sd["quantizer.input_proj.weight"], self.sd["quantizer.input_proj.bias"])
        if qc.get("codec_range", None) is not None:
            emb = torch.tanh(emb) * qc["codec_range"]
        x = emb.permute(0, 2, 1)                                  # [B,D,T]
        residual = x
        out = torch.zeros_like(x)
        all_idx, all_sub = [], []
        tt = x.shape[2]
        q0_ds = qc.get("q0_ds_ratio", 1) > 1                      # ddp_core_vq.py:354-356
        for i in range(n_q):
            e = self.embed[i]
            quant_in = residual
            if q0_ds and i == 0:                                  # :395-397 (mode "nearest"; the reference halves whatever the ratio is)
                quant_in = F.interpolate(quant_in, size=[tt // 2])
            xi = quant_in.permute(0, 2, 1)                        # rearrange b d n -> b n d  (:318)
            shape = xi.shape
            flat = xi.reshape(-1, shape[-1])                      # preprocess (... d -> (...) d)
            embed_t = e.t()
            dist = -(flat.pow(2).sum(1, keepdim=True) - 2 * flat @ embed_t + embed_t.pow(2).sum(0, keepdim=True))  # :180-187
            ind = dist.max(dim=-1).indices
            ind = ind.view(*shape[:-1])
            quant = F.embedding(ind, e).permute(0, 2, 1)          # :190-192, :323
            if q0_ds and i == 0:                                  # :404-406
                quant = F.interpolate(quant, size=[tt])
                ind = F.interpolate(ind.unsqueeze(1).float(), size=[tt]).squeeze(1).long()
            residual = residual - quant
            out = out + quant
            all_idx.append(ind)
            all_sub.append(quant)
        quantized = out.permute(0, 2, 1)
        if "quantizer.output_proj.weight" in self.sd:             # :92-94
            quantized = F.linear(quantized, self.sd["quantizer.output_proj.weight"], self.sd["quantizer.output_proj.bias"])
        return quantized, torch.stack(all_idx), torch.stack(all_sub)

    @torch.no_grad()
    def rvq_decode(self, codes: torch.Tensor) -> torch.Tensor:
        """DRVQ.decode ddp_core_vq.py:442-453: codes [n_q,B,Tf] -> [B,D,Tf]."""
        out = torch.tensor(0.0)
        for i, ind in enumerate(codes):
            out = out + F.embedding(ind, self.embed[i]).permute(0, 2, 1)
        if "quantizer.output_proj.weight" in self.sd:             # CostumeQuantizer.decode :114-119
            out = F.linear(out.transpose(1, 2), self.sd["quantizer.output_proj.weight"], self.sd["quantizer.output_proj.bias"]).transpose(1, 2)
        return out

    # -- Encodec.inference* --------------------------------------------------------------------
    @torch.no_grad()
    def encode_frame(self, speech: torch.Tensor):
        """Encodec._encode_frame codec_basic.py:361-380. speech [B,1,T] -> emb [B,Tf,D], scale [B,1]|None."""
        x = speech
        scale = None
        if self.audio_normalize:
            mono = x.mean(dim=1, keepdim=True)
            volume = mono.pow(2).mean(dim=2, keepdim=True).sqrt()
            scale = 1e-8 + volume
            x = x / scale
            scale = scale.view(-1, 1)
        return self.encoder(x), scale

    @staticmethod
    def linear_overlap_add(frames, stride):
        """_linear_overlap_add codec_basic.py:77-116."""
        total = stride * (len(frames) - 1) + frames[-1].shape[-1]
        flen = frames[0].shape[-1]
        t = torch.linspace(0, 1, flen + 2, dtype=frames[0].dtype)[1:-1]
        weight = 0.5 - (t - 0.5).abs()
        sum_w = torch.zeros(total, dtype=frames[0].dtype)
        out = torch.zeros(*frames[0].shape[:-1], total, dtype=frames[0].dtype)
        off = 0
        for f in frames:
            n = f.shape[-1]
            out[..., off:off + n] += weight[:n] * f
            sum_w[off:off + n] += weight[:n]
            off += stride
        return out / sum_w

    @torch.no_grad()
    def inference_segmented(self, speech: torch.Tensor, bit_width=None, use_scale=True, need_recon=True):
        """Encodec.inference with model_conf.segment_dur set: _encode frames (codec_basic.py:334-359), per-frame RVQ, per-frame
        _decode_frame, _linear_overlap_add (:382-396), trim to the input length (:711)."""
        if speech.dim() == 2:
            speech = speech.unsqueeze(1)
        m = self.cfg.get("model_conf", {})
        sr = int(m.get("target_sample_hz", 24000))                   # Encodec.__init__ defaults, codec_basic.py:132-141
        seg = int(m.get("segment_dur", 1.0) * sr)
        ov = 0.01 if m.get("overlap_ratio", 0.01) is None else m.get("overlap_ratio", 0.01)
        stride = max(1, int((1 - ov) * seg))
        T = speech.shape[-1]
        idxs, embs, subs_all, recons, encs, scales = [], [], [], [], [], []
        for off in range(0, T, stride):
            frame = speech[:, :, off:off + seg]
            emb, scale = self.encode_frame(frame)
            quant, idx, subs = self._quantise(emb, bit_width)
            idxs.append(idx); embs.append((quant, scale if use_scale else None)); subs_all.append(subs)
            encs.append(emb); scales.append(scale)
            if need_recon:
                r = self.decoder(quant)
                if use_scale and scale is not None:
                    r = r * scale.view(-1, 1, 1)
                recons.append(r)
        recon = self.linear_overlap_add(recons, stride)[:, :, :T] if need_recon else None
        return dict(code_indices=idxs, code_embeddings=embs, recon_speech=recon, sub_quants=subs_all,
                    encoder_out=encs, scale=scales)

    def _quantise(self, emb, bit_width):
        """the quantiser call of Encodec.inference, or model_conf.bypass_quantizer's stand-ins (codec_basic.py:700-705)"""
        if self.cfg.get("model_conf", {}).get("bypass_quantizer", False):
            return emb, torch.zeros(emb.shape[0], emb.shape[1], dtype=torch.long), torch.zeros_like(emb)
        return self.rvq_forward(emb, self.n_q_for(bit_width))

    @torch.no_grad()
    def inference(self, speech: torch.Tensor, bit_width=None, use_scale=True, need_recon=True):
        """Encodec.inference codec_basic.py:670-718 (one frame when segment_dur is null)."""
        if self.cfg.get("model_conf", {}).get("segment_dur", 1.0) is not None:
            return self.inference_segmented(speech, bit_width, use_scale, need_recon)
        if speech.dim() == 2:
            speech = speech.unsqueeze(1)
        emb, scale = self.encode_frame(speech)
        quant, idx, subs = self._quantise(emb, bit_width)
        recon = None
        if need_recon:
            recon = self.decoder(quant)
            if use_scale and scale is not None:
                recon = recon * scale.view(-1, 1, 1)                # _decode_frame :406-407
            recon = recon[:, :, :speech.shape[-1]]
        return dict(code_indices=[idx], code_embeddings=[(quant, scale if use_scale else None)],
                    recon_speech=recon, sub_quants=[subs], encoder_out=emb, scale=scale)

    @torch.no_grad()
    def decode_codes(self, token_idx: torch.Tensor):
        """Encodec.inference_decoding codec_basic.py:766-802: token_idx [B,Tf,n_q] -> wav [B,1,Tf*hop]."""
        codes = token_idx.permute(2, 0, 1)
        emb = self.rvq_decode(codes).transpose(1, 2)
        return self.decoder(emb), emb

    @torch.no_grad()
    def decode_emb(self, emb: torch.Tensor):
        """Encodec.inference_decoding_emb codec_basic.py:804-836."""
        return self.decoder(emb)
