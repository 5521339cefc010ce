"""Host side of the MI355X LauraTTS engine and the drop-in for the inference surface of the reference's ``LauraGenModel``
(funcodec/models/audio_generation/laura_model.py): ``encode`` (:186-202), ``decode_codec`` (:501-548), ``cal_codec_emb``
(:296-333), ``syn_audio`` (:550-567), plus batch forms of each (the reference generates one utterance at a time, re-scoring the
whole prefix per token; the engine keeps a KV cache, decodes up to 16 prompts per call and samples on the device).

PyTorch is used for device memory, streams and checkpoint I/O only; every arithmetic operation happens inside
libfuncodec_amd.so (HIP, gfx950).  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .engine import EngineError, _on_device, _ptr
from .laura_config import LauraSpec, StackSpec, laura_spec_from_config


def _i32(vals: Sequence[int]):
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


def sampling_args(sampling: Union[bool, int, float]):
    """LauraGenModel.sampling_ids' `sampling` argument (laura_model.py:466-499) -> (mode, k, p) of fc_laura_decode_codec."""
    if isinstance(sampling, bool):
        return (1, 0, 0.0) if sampling else (0, 0, 0.0)
    if isinstance(sampling, int):
        return 2, int(sampling), 0.0
    if isinstance(sampling, float):
        return 3, 0, float(sampling)
    raise NotImplementedError(f"Not implemented for {type(sampling)} sampling")


class LauraEngine:
    """One engine per (device, checkpoint); calls are serialised by the caller like a torch module's forward."""

    max_batch = 16

    def __init__(self, spec: LauraSpec, device: "torch.device | str | int" = "cuda:0", max_positions: int = 2048):
        self.lib = _lib.load()
        self.spec = spec
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if dev.type != "cuda":
            raise EngineError(f"funcodec_amd runs on MI355X (gfx950) only; device={device!r} has no implementation "
                              "(there is deliberately no CPU fallback)")
        self.device = torch.device("cuda", dev.index if dev.index is not None else 0)
        if int(max_positions) % 4 or not 16 <= int(max_positions) <= 2048:
            raise EngineError(f"max_positions must be a multiple of 4 in [16, 2048], got {max_positions}")
        a = _lib.FcLauraArch()
        a.abi_version = _lib.FC_ABI_VERSION
        a.input_size, a.vocab_size = spec.input_size, spec.vocab_size
        a.codebook_size, a.codebook_dim, a.num_quantizers = spec.codebook_size, spec.codebook_dim, spec.num_quantizers
        a.predict_nq = spec.predict_nq
        a.pos_emb_split = int(spec.pos_emb_type == "split")
        a.bidirectional_inputs = int(spec.bidirectional_inputs)
        a.max_positions = int(max_positions)
        for name in ("text_encoder", "codec_lm", "codec_encoder"):
            s: StackSpec = getattr(spec, name)
            d = getattr(a, name)
            d.idim, d.d_model, d.heads, d.ff, d.layers = s.idim, s.d_model, s.heads, s.ff, s.layers
            d.act = {"relu": 1, "swish": 2}[s.act]
            d.embed_relu = int(s.embed_relu)
            d.norm_style = int(s.norm_names[0] == "norm1")
        self.max_positions = int(max_positions)
        h = C.c_void_p()
        self._check(self.lib.fc_laura_create(C.byref(a), self.device.index, C.byref(h)))
        self._h = h
        self._ws: Optional[torch.Tensor] = None
        self._ws_need: Dict[tuple, int] = {}
        self._side = None

    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(self.lib.fc_last_error().decode("utf-8", "replace"))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.fc_laura_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def expected_tensors(self) -> Dict[str, tuple]:
        out = {}
        name = C.c_char_p()
        dims = (C.c_int64 * 4)()
        for i in range(self.lib.fc_laura_num_weights(self._h)):
            nd = self.lib.fc_laura_weight_info(self._h, i, C.byref(name), dims)
            out[name.value.decode()] = tuple(int(dims[j]) for j in range(nd))
        return out

    @_on_device
    def load_state_dict(self, state: Dict[str, "torch.Tensor | np.ndarray"]) -> None:
        """Tolerant load like the reference's filter_state_dict (funcodec/torch_utils/load_pretrained_model.py:12-43): tensors
        outside the generation path (the model's training-time quantiser, codec_index_shift) are skipped; a MISSING tensor of the
        path is an error."""
        want = self.expected_tensors()
        for key, shape in want.items():
            if key not in state:
                raise EngineError(f"checkpoint is missing tensor {key} {shape}")
            t = torch.as_tensor(state[key]).detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise EngineError(f"shape mismatch for {key}: checkpoint {tuple(t.shape)} vs architecture {shape}")
            dims = (C.c_int64 * 4)(*t.shape)
            self._check(self.lib.fc_laura_set_weight(self._h, key.encode(), C.c_void_p(t.data_ptr()), dims, t.dim()))
        skipped = [k for k in state if k not in want]
        if skipped:
            logging.info("funcodec_amd: skipped %d LauraTTS checkpoint tensors outside the generation path (e.g. %s)", len(skipped), skipped[0])
        self._check(self.lib.fc_laura_finalize(self._h))

    _fallbacks_seen = 0

    @property
    def persistent_step_fallbacks(self) -> int:
        """Calls of this engine whose persistent decoding step timed out and which were re-run on the kernel chain."""
        return int(self.lib.fc_laura_persistent_step_fallbacks(self._h))

    def set_persistent_step(self, on: bool) -> bool:
        """Decoding step as one persistent launch (default) or as the chain of one kernel per Linear / attention; returns whether the
        persistent form is in effect (False also when this model / device cannot run it)."""
        return self.lib.fc_laura_set_persistent_step(self._h, int(bool(on))) == 1

    def _workspace(self, B: int, L: int, Cmax: int, max_length: int) -> torch.Tensor:
        key = (B, L, Cmax, max_length)
        need = self._ws_need.get(key)
        if need is None:
            need = self._ws_need[key] = int(self.lib.fc_laura_workspace_bytes(self._h, B, L, Cmax, max_length))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype) -> torch.Tensor:
        return torch.as_tensor(t).to(device=self.device, dtype=dtype).contiguous()

    # -- argument validation: the C side takes plain pointers and trusts these shapes ------------------------------------------------
    def _check_batch(self, what: str, B: int, cap: Optional[int] = None, **length_lists) -> None:
        if B < 1 or (cap is not None and B > cap):
            raise EngineError(f"{what}: batch of {B} utterances" + (f", the engine takes 1 .. {cap} per call" if cap else ""))
        for name, vals in length_lists.items():
            if vals is None:
                continue
            if len(vals) != B:
                raise EngineError(f"{what}: {name} has {len(vals)} entries for a batch of {B}")

    def _check_lens(self, what: str, name: str, vals: Sequence[int], upper: int, lower: int = 0) -> List[int]:
        out = [int(v) for v in vals]
        for v in out:
            if v < lower or v > upper:
                raise EngineError(f"{what}: {name} entry {v} outside [{lower}, {upper}]")
        return out

    def _check_text_outs(self, what: str, text_outs: torch.Tensor) -> None:
        if text_outs.dim() != 3 or text_outs.shape[-1] != self.spec.codebook_dim:
            raise EngineError(f"{what}: text_outs must be [B, L, {self.spec.codebook_dim}], got {tuple(text_outs.shape)}")

    # -- LauraGenModel.encode -------------------------------------------------------------------------------------------
    @_on_device
    def encode(self, text, text_lengths: Sequence[int]) -> torch.Tensor:
        """text: float [B, L, input_size] embeddings or int64 [B, L] token ids -> text_outs [B, L, codebook_dim]."""
        text = torch.as_tensor(text)
        ids = not text.is_floating_point()
        text = self._dev(text, torch.int64 if ids else torch.float32)
        if (ids and text.dim() != 2) or (not ids and (text.dim() != 3 or text.shape[-1] != self.spec.input_size)):
            raise EngineError(f"encode: text must be int64 [B, L] token ids or float [B, L, {self.spec.input_size}] embeddings, "
                              f"got {tuple(text.shape)}")
        B, L = text.shape[0], text.shape[1]
        text_lengths = list(text_lengths)
        self._check_batch("encode", B, text_lengths=text_lengths)
        text_lengths = self._check_lens("encode", "text_lengths", text_lengths, L, 1)
        out = torch.empty((B, L, self.spec.codebook_dim), dtype=torch.float32, device=self.device)
        ws = self._workspace(B, L, 0, 1)
        self._check(self.lib.fc_laura_encode(self._h, None if ids else _ptr(text), _ptr(text) if ids else None, _i32(text_lengths), B, L,
                                             _ptr(out), _ptr(ws), ws.numel(), self._stream()))
        return out

    # -- teacher-forced LM scores ---------------------------------------------------------------------------------------
    @_on_device
    def lm_logprobs(self, text_outs: torch.Tensor, text_lengths: Sequence[int], codec: Optional[torch.Tensor] = None,
                    codec_lengths: Optional[Sequence[int]] = None) -> torch.Tensor:
        """log-softmax of the LM output at every position of [<sos>, text, <task>, codec]: [B, Tseq, vocab]."""
        text_outs = self._dev(text_outs, torch.float32)
        self._check_text_outs("lm_logprobs", text_outs)
        B, L = text_outs.shape[0], text_outs.shape[1]
        text_lengths = list(text_lengths)
        self._check_batch("lm_logprobs", B, text_lengths=text_lengths)
        text_lengths = self._check_lens("lm_logprobs", "text_lengths", text_lengths, L, 1)
        Cmax = 0
        if codec is not None:
            codec = self._dev(codec, torch.int64)
            if codec.dim() != 3 or codec.shape[0] != B or codec.shape[2] != self.spec.predict_nq:
                raise EngineError(f"lm_logprobs: codec must be int64 [{B}, Cmax, {self.spec.predict_nq}], got {tuple(codec.shape)}")
            Cmax = codec.shape[1]
            if codec_lengths is None:
                raise EngineError("lm_logprobs: codec given without codec_lengths")
            self._check_batch("lm_logprobs", B, codec_lengths=list(codec_lengths))
        cl = self._check_lens("lm_logprobs", "codec_lengths", list(codec_lengths), Cmax) if codec is not None else [0] * B
        Tseq = max(int(t) + 2 + int(c) for t, c in zip(text_lengths, cl))
        logp = torch.empty((B, Tseq, self.spec.lm_vocab), dtype=torch.float32, device=self.device)
        ws = self._workspace(B, L, Cmax, 1)
        self._check(self.lib.fc_laura_lm_logprobs(self._h, _ptr(text_outs), _i32(text_lengths), B, L, _ptr(codec),
                                                  _i32(cl) if codec is not None else None, Cmax, _ptr(logp), Tseq, _ptr(ws), ws.numel(),
                                                  self._stream()))
        return logp

    # -- LauraGenModel.decode_codec ----------------------------------------------------------------------------------------
    @_on_device
    def decode_codec(self, text_outs: torch.Tensor, text_lengths: Sequence[int], max_length: int = 30 * 25,
                     sampling: Union[bool, int, float] = True, seed: int = 0, continual: Optional[torch.Tensor] = None,
                     continual_lengths: Optional[Sequence[int]] = None, forced: Optional[torch.Tensor] = None,
                     return_logp: bool = False):
        """Batch form of decode_codec: returns (tokens [B, Cmax + max_length, nq] int64, lengths list[int]) and, with
        return_logp, the per-step log-probabilities [B, max_length, vocab]."""
        text_outs = self._dev(text_outs, torch.float32)
        self._check_text_outs("decode_codec", text_outs)
        B, L = text_outs.shape[0], text_outs.shape[1]
        text_lengths = list(text_lengths)
        self._check_batch("decode_codec", B, cap=self.max_batch, text_lengths=text_lengths)
        text_lengths = self._check_lens("decode_codec", "text_lengths", text_lengths, L, 1)
        if int(max_length) < 1:
            raise EngineError(f"decode_codec: max_length {max_length} < 1")
        nq = self.spec.predict_nq
        Cmax = 0
        if continual is not None:
            continual = self._dev(continual, torch.int64)
            if continual.dim() != 3 or continual.shape[0] != B or continual.shape[2] != nq:
                raise EngineError(f"decode_codec: continual must be int64 [{B}, Cmax, {nq}], got {tuple(continual.shape)}")
            Cmax = continual.shape[1]
            if continual_lengths is None:
                raise EngineError("decode_codec: continual given without continual_lengths")
            continual_lengths = list(continual_lengths)
            self._check_batch("decode_codec", B, continual_lengths=continual_lengths)
            continual_lengths = self._check_lens("decode_codec", "continual_lengths", continual_lengths, Cmax)
        mode, k, p = sampling_args(sampling)
        if forced is not None:
            forced = self._dev(forced, torch.int64)
            if tuple(forced.shape) != (B, int(max_length), nq):
                raise EngineError(f"decode_codec: forced must be int64 [{B}, {int(max_length)}, {nq}], got {tuple(forced.shape)}")
        tokens = torch.zeros((B, Cmax + max_length, nq), dtype=torch.int64, device=self.device)
        logp = torch.zeros((B, max_length, self.spec.lm_vocab), dtype=torch.float32, device=self.device) if return_logp else None
        out_lens = (C.c_int32 * B)()
        ws = self._workspace(B, L, Cmax, max_length)
        # the decoding loop replays a captured HIP graph of one step; the legacy default stream cannot be captured, so a caller on it
        # is moved to a stream of the engine's own (ordered after / before the caller's stream; the call synchronises at its end anyway)
        cur = torch.cuda.current_stream(self.device)
        run = cur
        if cur.cuda_stream == 0:
            if self._side is None:
                self._side = torch.cuda.Stream(self.device)
            run = self._side
            run.wait_stream(cur)
        self._check(self.lib.fc_laura_decode_codec(self._h, _ptr(text_outs), _i32(text_lengths), B, L, _ptr(continual),
                                                   _i32(continual_lengths) if continual is not None else None, Cmax, int(max_length),
                                                   mode, k, p, int(seed) & (2 ** 64 - 1), _ptr(forced), _ptr(tokens), out_lens, _ptr(logp),
                                                   _ptr(ws), ws.numel(), C.c_void_p(run.cuda_stream)))
        if run is not cur:
            cur.wait_stream(run)
        nfb = int(self.lib.fc_laura_persistent_step_fallbacks(self._h))
        if nfb != self._fallbacks_seen:            # the call succeeded on the kernel chain after a hand-off timeout: said, not hidden
            self._fallbacks_seen = nfb
            import warnings
            warnings.warn("decode_codec: the persistent decoding step timed out at a hand-off (CUs held by another stream / process?); this call "
                          "was re-run on the kernel chain and later calls use it until set_persistent_step(True)", RuntimeWarning)
        lens = [int(v) for v in out_lens]
        return (tokens, lens, logp) if return_logp else (tokens, lens)

    # -- LauraGenModel.cal_codec_emb on one-hot probabilities (syn_audio) ---------------------------------------------------
    @_on_device
    def codec_emb(self, text_outs: torch.Tensor, text_lengths: Sequence[int], codec: torch.Tensor, codec_lengths: Sequence[int]) -> torch.Tensor:
        """codec int64 [B, Cmax, >= predict_nq] -> dense codec embeddings [B, Cmax, codebook_dim] (rows past each length zero)."""
        text_outs = self._dev(text_outs, torch.float32)
        codec = self._dev(codec, torch.int64)
        self._check_text_outs("codec_emb", text_outs)
        B, L = text_outs.shape[0], text_outs.shape[1]
        if codec.dim() != 3 or codec.shape[0] != B or codec.shape[2] < self.spec.predict_nq:
            raise EngineError(f"codec_emb: codec must be int64 [{B}, Cmax, >= {self.spec.predict_nq}], got {tuple(codec.shape)}")
        Cmax, cols = codec.shape[1], codec.shape[2]
        text_lengths, codec_lengths = list(text_lengths), list(codec_lengths)
        self._check_batch("codec_emb", B, text_lengths=text_lengths, codec_lengths=codec_lengths)
        text_lengths = self._check_lens("codec_emb", "text_lengths", text_lengths, L, 1)
        codec_lengths = self._check_lens("codec_emb", "codec_lengths", codec_lengths, Cmax)
        emb = torch.empty((B, Cmax, self.spec.codebook_dim), dtype=torch.float32, device=self.device)
        ws = self._workspace(B, L, Cmax, 1)
        self._check(self.lib.fc_laura_codec_emb(self._h, _ptr(text_outs), _i32(text_lengths), B, L, _ptr(codec), cols, _i32(codec_lengths),
                                                Cmax, _ptr(emb), _ptr(ws), ws.numel(), self._stream()))
        return emb

    # -- per-op entry point (tests) -----------------------------------------------------------------------------------------------
    @_on_device
    def linear(self, name: str, x: torch.Tensor, step_form: bool = False) -> torch.Tensor:
        x = self._dev(x, torch.float32)
        B, T, cin = x.shape
        want = self.expected_tensors()
        cout = want[name + ".weight"][0] if name + ".weight" in want else 3 * cin
        y = torch.empty((B, T, cout), dtype=torch.float32, device=self.device)
        need = 4 * B * (T + 8) * (cin + cout) * 2 + (1 << 20)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ws = self._ws
        self._check(self.lib.fc_laura_linear(self._h, name.encode(), _ptr(x), B, T, int(step_form), _ptr(y), _ptr(ws), ws.numel(), self._stream()))
        return y


class LauraGenMI355X:
    """Drop-in for the inference surface of ``LauraGenModel``: same method names, argument meaning and return values
    (batch-1 tensors in, tensors out) as funcodec/models/audio_generation/laura_model.py, plus ``*_batch`` forms."""

    def __init__(self, spec: LauraSpec, device="cuda:0", max_positions: int = 2048):
        self.spec = spec
        self.engine = LauraEngine(spec, device, max_positions)
        self.device = self.engine.device
        self.vocab_size = spec.vocab_size
        self.token_list = spec.token_list
        self.predict_nq = spec.predict_nq
        self.codebook_size = spec.codebook_size
        self.codebook_dim = spec.codebook_dim
        self.sos_eos, self.task_id = 0, 1
        self.training = False
        self._seed_counter = 0

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def load_state_dict(self, state, strict: bool = False):
        self.engine.load_state_dict(state)

    # Text2Audio.tokenize_text reads model.token_embedding(token_idx) (bin/text2audio_inference.py:112); the lookup happens inside
    # fc_laura_encode, so the "embedding" of the drop-in is the id tensor itself
    def token_embedding(self, token_idx: torch.Tensor) -> torch.Tensor:
        return torch.as_tensor(token_idx).long()

    @torch.no_grad()
    def encode(self, text: torch.Tensor, text_lengths: torch.Tensor):
        lens = [int(v) for v in torch.as_tensor(text_lengths).reshape(-1)]
        return self.engine.encode(text, lens), torch.as_tensor(lens, dtype=torch.int64, device=self.device)

    def _next_seed(self) -> int:
        # the reference draws from torch's global generator; a generation here is a function of (torch seed, call index)
        self._seed_counter += 1
        return (int(torch.initial_seed()) * 1000003 + self._seed_counter) & (2 ** 64 - 1)

    @torch.no_grad()
    def decode_codec(self, text: torch.Tensor, text_lengths: torch.Tensor, max_length: int = 30 * 25,
                     sampling: Union[bool, int, float] = True, beam_size: int = 1, continual: List = None, seed: Optional[int] = None) -> torch.Tensor:
        """One utterance, like the reference: text [1, L, D] -> tokens [1, T, predict_nq]."""
        cont = None
        cl = None
        if continual is not None and len(continual) > 0:
            cont = torch.as_tensor(continual, dtype=torch.int64).reshape(1, -1, self.predict_nq)
            cl = [cont.shape[1]]
        tokens, lens = self.engine.decode_codec(text, [int(torch.as_tensor(text_lengths).reshape(-1)[0])], max_length, sampling,
                                                self._next_seed() if seed is None else seed, cont, cl)
        return tokens[:, : lens[0]]

    @torch.no_grad()
    def decode_codec_batch(self, text: torch.Tensor, text_lengths: Sequence[int], max_length: int = 30 * 25,
                           sampling: Union[bool, int, float] = True, continual: Optional[torch.Tensor] = None,
                           continual_lengths: Optional[Sequence[int]] = None, seed: Optional[int] = None):
        return self.engine.decode_codec(text, text_lengths, max_length, sampling, self._next_seed() if seed is None else seed, continual,
                                        continual_lengths)

    @torch.no_grad()
    def cal_codec_emb_batch(self, text, text_lengths, codec, codec_lengths):
        return self.engine.codec_emb(text, text_lengths, codec, codec_lengths)

    @torch.no_grad()
    def syn_audio(self, codec: torch.Tensor, text: torch.Tensor, text_lengths: torch.Tensor, codec_model, continual_length=None):
        """laura_model.py:550-567: codec [1, T, nq] -> waveform through the codec model's decode_emb."""
        tl = [int(torch.as_tensor(text_lengths).reshape(-1)[0])]
        emb = self.engine.codec_emb(text, tl, codec, [codec.shape[1]])
        _, _, recon_wav, _ = codec_model(emb[:, continual_length:], run_mod="decode_emb")
        return recon_wav
