"""Drop-in for the inference surface of the reference's ``Encodec`` model
(funcodec/models/codec_basic.py:670-836) on top of :class:`funcodec_amd.engine.CodecEngine`.

Same method names, argument meaning and return dictionaries as the reference, so that
``Speech2Token`` (and LauraTTS's use of it, funcodec/bin/text2audio_inference.py:85-94,157,180-190)
can switch without touching call sites.
"""
from __future__ import annotations

import types
from typing import Dict, Optional

import torch

from .config import ArchSpec
from .engine import CodecEngine


class EncodecMI355X:
    def __init__(self, arch: ArchSpec, device="cuda:0"):
        self.arch = arch
        self.engine = CodecEngine(arch, device)
        self.device = self.engine.device
        # attributes other reference code reaches into
        self.quantizer = types.SimpleNamespace(
            sampling_rate=arch.quantizer_sampling_rate,
            encoder_hop_length=arch.encoder_hop_length,
            codebook_size=arch.codebook_size,
            code_dim=arch.dimension,
            get_num_quantizers_for_bandwidth=lambda sr, bw=None: arch.num_quantizers_for_bandwidth(bw),
        )
        self.sample_rate = arch.sample_rate
        self.training = False

    # nn.Module-ish no-ops so callers written against the reference keep working
    def eval(self):
        return self

    def to(self, *args, **kwargs):
        return self

    def load_state_dict(self, state, strict: bool = False):
        self.engine.load_state_dict(state)

    # -- helpers -------------------------------------------------------------------------------
    def _as_bct(self, speech: torch.Tensor) -> torch.Tensor:
        if speech.dim() == 2:
            speech = speech.unsqueeze(1)
        assert speech.dim() == 3, "speech must be [B,T] or [B,C,T]"          # codec_basic.py:342
        assert 0 < speech.shape[1] <= 2                                       # codec_basic.py:344
        if speech.shape[1] != self.engine.channels:      # the reference's first conv would raise on the channel mismatch
            raise ValueError(f"this model takes {self.engine.channels}-channel audio (config input_size), got {speech.shape[1]} channels")
        return speech

    # -- Encodec._encode / _decode with segment_dur set (codec_basic.py:334-359,382-396,77-116) --
    def _inference_segmented(self, wav: torch.Tensor, n_q: int, need_recon: bool, use_scale: bool, bypass: bool = False):
        """Every frame is an independent utterance for the engine (own volume scale, GroupNorm statistics, LSTM state), so
        all frames of equal length go through ONE engine call as extra batch rows; the ragged tail frames follow.  Like the
        reference's _decode, every frame is decoded to its FULL length ceil(len/hop)*hop (longer than the segment when the
        segment length is not a multiple of the hop), the triangle window is sized from the first decoded frame, and the
        overlap-add result is trimmed to the input length last (fc_overlap_add)."""
        B, T = wav.shape[0], wav.shape[-1]
        seg, stride = self.arch.segment_length, self.arch.segment_stride
        offsets = list(range(0, T, stride))
        lens = [min(seg, T - o) for o in offsets]
        results = [None] * len(offsets)
        by_len = {}
        for i, n in enumerate(lens):
            by_len.setdefault(n, []).append(i)
        for n, ids in by_len.items():
            stack = torch.cat([wav[..., offsets[i]:offsets[i] + n] for i in ids], 0).contiguous()   # [len(ids)*B, (C,) n]
            r = self._encode_or_bypass(stack, n_q, bypass)
            rec = None
            if need_recon:
                rec = self.engine.decode_emb(r["quantized"], r["scale"] if use_scale else None)   # untrimmed: Tf*hop samples
            for j, i in enumerate(ids):
                sl = slice(j * B, (j + 1) * B)
                results[i] = dict(codes=r["codes"][sl] if bypass else r["codes"][:, sl], quantized=r["quantized"][sl],
                                  sub_quants=r["sub_quants"][sl] if bypass else r["sub_quants"][:, sl],
                                  scale=r["scale"][sl] if r.get("scale") is not None else None,
                                  recon=rec[sl] if need_recon else None)
        recon = None
        if need_recon:
            recon = self.engine.overlap_add([r["recon"] for r in results], stride, out_len=T)
        return dict(recon_speech=recon, code_indices=[r["codes"] for r in results],
                    code_embeddings=[(r["quantized"], r["scale"] if use_scale else None) for r in results],
                    sub_quants=[r["sub_quants"] for r in results])

    def _encode_or_bypass(self, wav: torch.Tensor, n_q: int, bypass: bool):
        """engine.encode, or with model_conf.bypass_quantizer (codec_basic.py:700-701, Encodec.inference only) the encoder output in place of
        the quantised embeddings, zero indices [B, Tf] and zero sub_quants like the reference builds them."""
        if not bypass:
            return self.engine.encode(wav, n_q)
        r = self.engine.encode(wav, 1, want_sub_quants=False, want_enc_out=True)     # the quantiser's stage is computed and dropped
        emb = r["enc_out"]
        return dict(codes=torch.zeros(emb.shape[0], emb.shape[1], dtype=torch.long, device=emb.device), quantized=emb,
                    sub_quants=torch.zeros_like(emb), scale=r["scale"])

    # -- Encodec.inference (codec_basic.py:670-718) ------------------------------------------
    @torch.no_grad()
    def inference(self, speech: torch.Tensor, need_recon: bool = True, bit_width: int = None,
                  use_scale: bool = True, _quantise_always: bool = False) -> Dict[str, torch.Tensor]:
        speech = self._as_bct(speech)
        bypass = self.arch.bypass_quantizer and not _quantise_always
        n_q = self.arch.num_quantizers_for_bandwidth(bit_width)
        wav = speech[:, 0, :] if self.engine.channels == 1 else speech
        if self.arch.segment_length is not None:
            wav = wav.to(self.device, torch.float32)
            return self._inference_segmented(wav, n_q, need_recon, use_scale, bypass)
        if bypass:
            r = self._encode_or_bypass(wav, n_q, True)
            recon = None
            if need_recon:             # _decode_frame on the encoder output, trimmed like recon[:, :, :T] (:711)
                T = wav.shape[-1]
                recon = self.engine.decode_emb(r["quantized"], r["scale"] if use_scale else None,
                                               out_len=min(T, self.engine.decoded_samples(r["quantized"].shape[1])))
        elif need_recon:
            r = self.engine.encode_decode(wav, n_q, use_scale=use_scale)
            recon = r["recon"]
        else:
            r = self.engine.encode(wav, n_q)
            recon = None
        scale = r["scale"] if use_scale else None
        return dict(recon_speech=recon, code_indices=[r["codes"]], code_embeddings=[(r["quantized"], scale)],
                    sub_quants=[r["sub_quants"]])

    # -- Encodec.inference_encoding (codec_basic.py:720-764) ---------------------------------
    @torch.no_grad()
    def inference_encoding(self, speech: torch.Tensor, need_recon: bool = False, bit_width: int = None,
                           use_scale: bool = True) -> Dict[str, torch.Tensor]:
        # (model_conf.bypass_quantizer does not reach this entry point in the reference: it quantises, :748-750)
        return self.inference(speech, need_recon=need_recon, bit_width=bit_width, use_scale=use_scale, _quantise_always=True)

    # -- Encodec.inference_decoding (codec_basic.py:766-802) ---------------------------------
    @torch.no_grad()
    def inference_decoding(self, token_idx: torch.Tensor, need_recon: bool = True, bit_width: int = None,
                           use_scale: bool = True) -> Dict[str, torch.Tensor]:
        recon, emb = self.engine.decode_codes(token_idx)
        if self.arch.segment_length is not None and need_recon:      # _decode: one frame through the overlap-add (codec_basic.py:396)
            recon = self.engine.overlap_add([recon], self.arch.segment_stride or 1)
        return dict(recon_speech=recon if need_recon else None, code_indices=None,
                    code_embeddings=[(emb, None)], sub_quants=None)

    # -- Encodec.inference_decoding_emb (codec_basic.py:804-836) -----------------------------
    @torch.no_grad()
    def inference_decoding_emb(self, token_idx: torch.Tensor, need_recon: bool = True, bit_width: int = None,
                               use_scale: bool = True) -> Dict[str, torch.Tensor]:
        recon = self.engine.decode_emb(token_idx) if need_recon else None
        if self.arch.segment_length is not None and recon is not None:
            recon = self.engine.overlap_add([recon], self.arch.segment_stride or 1)
        return dict(recon_speech=recon, code_indices=None, code_embeddings=[(token_idx, None)], sub_quants=None)
