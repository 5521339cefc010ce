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
    @staticmethod
    def _as_b1t(speech: torch.Tensor) -> torch.Tensor:
        if speech.dim() == 2:
            speech = speech.unsqueeze(1)
        assert speech.dim() == 3, "speech must be [B,T] or [B,C,T]"          # codec_basic.py:342
        assert 0 < speech.shape[1] <= 2                                       # codec_basic.py:344
        if speech.shape[1] != 1:
            raise NotImplementedError("stereo input is outside the MI355X hot-path scope (SURVEY.md §8)")
        return speech

    # -- Encodec.inference (codec_basic.py:670-718) ------------------------------------------
    @torch.no_grad()
    def inference(self, speech: torch.Tensor, need_recon: bool = True, bit_width: int = None,
                  use_scale: bool = True) -> Dict[str, torch.Tensor]:
        speech = self._as_b1t(speech)
        n_q = self.arch.num_quantizers_for_bandwidth(bit_width)
        wav = speech[:, 0, :]
        if need_recon:
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
        return self.inference(speech, need_recon=need_recon, bit_width=bit_width, use_scale=use_scale)

    # -- Encodec.inference_decoding (codec_basic.py:766-802) ---------------------------------
    @torch.no_grad()
    def inference_decoding(self, token_idx: torch.Tensor, need_recon: bool = True, bit_width: int = None,
                           use_scale: bool = True) -> Dict[str, torch.Tensor]:
        recon, emb = self.engine.decode_codes(token_idx)
        return dict(recon_speech=recon if need_recon else None, code_indices=None,
                    code_embeddings=[(emb, None)], sub_quants=None)

    # -- Encodec.inference_decoding_emb (codec_basic.py:804-836) -----------------------------
    @torch.no_grad()
    def inference_decoding_emb(self, token_idx: torch.Tensor, need_recon: bool = True, bit_width: int = None,
                               use_scale: bool = True) -> Dict[str, torch.Tensor]:
        recon = self.engine.decode_emb(token_idx) if need_recon else None
        return dict(recon_speech=recon, code_indices=None, code_embeddings=[(token_idx, None)], sub_quants=None)
