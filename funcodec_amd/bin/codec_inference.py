"""``Speech2Token`` -- same constructor, call signature and 4-tuple as the reference's
funcodec/bin/codec_inference.py:41-150, running on the MI355X engine.

    from funcodec_amd.bin.codec_inference import Speech2Token
    s2t = Speech2Token("config.yaml", "model.pth", device="cuda")
    code_indices, code_embeddings, recon_speech, sub_quants = s2t(wav)        # run_mod="inference"
"""
from __future__ import annotations

import logging
import math
from pathlib import Path
from typing import Any, Optional, Union

import numpy as np
import torch

from ..config import arch_from_config
from ..model import EncodecMI355X


def build_model_from_file(config_file, model_file, device="cuda"):
    """Counterpart of GANSpeechCodecTask.build_model_from_file (funcodec/tasks/abs_task.py:1895-1947):
    yaml -> architecture, torch.load(model.pth) -> tolerant state_dict load."""
    import argparse
    import yaml
    with open(config_file, "rt", encoding="utf-8") as f:
        args = yaml.safe_load(f)
    arch = arch_from_config(args)
    model = EncodecMI355X(arch, device=device)
    if model_file is not None:
        state = torch.load(model_file, map_location="cpu")
        model.load_state_dict(state)
    return model, argparse.Namespace(**args)


class Speech2Token:
    """Speech2Token class (drop-in for funcodec.bin.codec_inference.Speech2Token)."""

    def __init__(
            self,
            config_file: Union[Path, str] = None,
            model_file: Union[Path, str] = None,
            device: str = "cuda",
            batch_size: int = 1,
            dtype: str = "float32",
            streaming: bool = False,
            sampling_rate: int = 24_000,
            bit_width: int = 24_000,
    ):
        if dtype != "float32":
            raise NotImplementedError("only dtype=float32 is supported (index exactness, SURVEY.md §7-3)")
        if device == "cpu":
            raise RuntimeError("funcodec_amd.Speech2Token runs on MI355X only; use the reference for device='cpu'")
        model, model_args = build_model_from_file(config_file, model_file, device)
        self.model = model
        self.model_args = model_args
        self.device = device
        self.dtype = dtype
        self.already_stat_flops = False

    @torch.no_grad()
    def __call__(
            self,
            speech: Union[torch.Tensor, np.ndarray],
            ppg: Optional[Union[torch.Tensor, np.ndarray]] = None,
            need_recon: bool = True,
            bit_width: int = None,
            use_scale: bool = True,
            run_mod: str = "inference",
    ):
        """Returns (code_indices, code_embeddings, recon_speech, sub_quants) like the reference (:86-134)."""
        if ppg is not None:
            raise NotImplementedError("ppg conditioning (codec_semantic_aug) is outside the hot-path scope")
        if isinstance(speech, np.ndarray):
            speech = torch.from_numpy(speech)
        speech = speech.to(self.model.device)
        if run_mod == "inference":
            ret = self.model.inference(speech, need_recon=need_recon, bit_width=bit_width, use_scale=use_scale)
        elif run_mod == "encode":
            ret = self.model.inference_encoding(speech, need_recon=False, bit_width=bit_width)
        elif run_mod == "decode_emb":
            ret = self.model.inference_decoding_emb(speech)
        else:
            q = self.model.quantizer
            bit_per_quant = (q.sampling_rate // q.encoder_hop_length) * int(math.log2(q.codebook_size))
            nq = None
            if bit_width is not None:
                nq = int(max(bit_width // bit_per_quant, 1))
            speech = speech[:, :, :nq]
            logging.info("use %d quantizers.", speech.shape[-1])
            ret = self.model.inference_decoding(speech)
        return (ret["code_indices"], ret["code_embeddings"], ret["recon_speech"], ret["sub_quants"])

    @staticmethod
    def from_pretrained(model_tag: Optional[str] = None, **kwargs: Optional[Any]):
        return Speech2Token(**kwargs)


class Token2Speech:
    """Convenience wrapper (the reference has no such class; decoding is Speech2Token(run_mod='decode'))."""

    def __init__(self, config_file=None, model_file=None, device: str = "cuda", speech2token: Speech2Token = None):
        self.s2t = speech2token or Speech2Token(config_file, model_file, device=device)

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor, bit_width: int = None) -> torch.Tensor:
        """tokens [B,Tf,n_q] int64 -> waveform [B,1,Tf*hop]"""
        return self.s2t(tokens, bit_width=bit_width, run_mod="decode")[2]
