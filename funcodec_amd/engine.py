"""Host side of the MI355X engine: owns the engine handle, checkpoint ingestion and the device
workspace.  PyTorch is used only for device memory, streams and checkpoint I/O; every arithmetic
operation of the hot path happens inside libfuncodec_amd.so (HIP, gfx950).
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .config import ArchSpec


class EngineError(RuntimeError):
    pass


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _on_device(fn):
    """Run a method with the engine's GPU as the thread's current device (kernels are launched on that device's stream;
    a caller whose current device is another GPU must not have to know)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        if not torch.cuda.is_available():          # GPU-less box: let the call reach the library's own loud error
            return fn(self, *a, **kw)
        with torch.cuda.device(self.device):
            return fn(self, *a, **kw)
    return wrapper


class CodecEngine:
    """One engine per (device, checkpoint).  Calls on one engine are serialised by the caller,
    like a torch module's forward."""

    #: utterances processed per engine call.  Every op of the path is per-utterance, so results do not depend on it
    #: (tests/test_gpu_parity.py::test_full_size_determinism_and_batch_independence); it bounds the workspace
    #: (~0.57 GB per 10 s utterance for ds640) and keeps the persistent LSTM kernel (B <= 32) on its fast path.
    micro_batch = 16

    def __init__(self, arch: ArchSpec, device: "torch.device | str | int" = "cuda:0"):
        self.lib = _lib.load()
        self.arch = arch
        if arch.lstm_layers > 0 and arch.bottleneck_channels == 512:
            # H = 512: the persistent LSTM advances two 16-utterance batch tiles side by side (128 workgroups each), so 32 utterances
            # per call cost the recurrence what 16 do
            self.micro_batch = 32
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if dev.type != "cuda":
            raise EngineError(
                f"funcodec_amd runs on MI355X (gfx950) only; device={device!r} has no implementation "
                "(there is deliberately no CPU fallback)")
        self.device = torch.device("cuda", dev.index if dev.index is not None else 0)
        a = _lib.FcArch()
        a.abi_version = _lib.FC_ABI_VERSION
        a.sample_rate = arch.sample_rate
        a.audio_normalize = int(arch.audio_normalize)
        a.n_filters = arch.n_filters
        a.dimension = arch.dimension
        a.n_ratios = len(arch.ratios)
        for i, r in enumerate(arch.ratios):
            a.ratios[i] = int(r)
        a.kernel_size = arch.kernel_size
        a.last_kernel_size = arch.last_kernel_size
        a.residual_kernel_size = arch.residual_kernel_size
        a.compress = arch.compress
        a.lstm_layers = arch.lstm_layers
        a.lstm_skip = int(arch.lstm_skip)
        a.elu_alpha = arch.elu_alpha
        a.gn_eps = arch.gn_eps
        a.codebook_size = arch.codebook_size
        a.num_quantizers = arch.num_quantizers
        a.norm_type = {"time_group_norm": 0, "weight_norm": 1, "none": 2}[arch.norm]
        a.causal = int(arch.causal)
        a.n_residual_layers = arch.n_residual_layers
        a.dilation_base = arch.dilation_base
        a.model_type = {"encodec": 0, "freq_codec": 1}[arch.model_type]
        a.input_channels = arch.input_channels
        # audio channels of the time-domain codec (stereo: wav / recon tensors are [B,2,T]); FreqCodec is mono
        self.channels = 2 if (arch.model_type == "encodec" and arch.input_channels == 2) else 1
        a.n_fft = arch.n_fft
        a.stft_hop = arch.stft_hop
        for i, r in enumerate(arch.ratios_f):
            a.ratios_f[i] = int(r)
        a.enc_conv_group_ratio, a.dec_conv_group_ratio = arch.enc_conv_group_ratio, arch.dec_conv_group_ratio
        a.dec_tr_conv_group_ratio = arch.dec_tr_conv_group_ratio
        a.codec_dim = arch.codebook_dim if arch.codebook_dim != arch.dimension else 0
        a.codec_range = float(arch.codec_range or 0.0)
        a.q0_ds_ratio = int(arch.q0_ds_ratio)
        h = C.c_void_p()
        self._check(self.lib.fc_engine_create(C.byref(a), self.device.index, C.byref(h)))
        self._h = h
        self._ws: Optional[torch.Tensor] = None
        self._ws_need: Dict[tuple, int] = {}
        self._finalized = False

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(self.lib.fc_last_error().decode("utf-8", "replace"))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.fc_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def expected_tensors(self) -> Dict[str, tuple]:
        out = {}
        name = C.c_char_p()
        dims = (C.c_int64 * 4)()
        for i in range(self.lib.fc_engine_num_weights(self._h)):
            nd = self.lib.fc_engine_weight_info(self._h, i, C.byref(name), dims)
            out[name.value.decode()] = tuple(int(dims[j]) for j in range(nd))
        return out

    # -- checkpoint ----------------------------------------------------------------------------
    @_on_device
    def load_state_dict(self, state: Dict[str, "torch.Tensor | np.ndarray"]) -> None:
        """Tolerant load, like the reference's filter_state_dict
        (funcodec/torch_utils/load_pretrained_model.py:12-43): unknown keys (discriminator.*,
        mel_spec_transforms.*, EMA buffers) are skipped; a MISSING hot-path tensor is an error
        because, unlike the reference, we cannot run on random init silently."""
        want = self.expected_tensors()
        state = dict(state)
        # `use_ddp: false` checkpoints store one codebook per layer (core_vq.py:147-150)
        if "quantizer.rq.model.embed" not in state:
            per = []
            i = 0
            while f"quantizer.rq.model.layers.{i}._codebook.embed" in state:
                per.append(torch.as_tensor(state[f"quantizer.rq.model.layers.{i}._codebook.embed"]))
                i += 1
            if per:
                state["quantizer.rq.model.embed"] = torch.stack(per)
        inited = state.get("quantizer.rq.model.inited", None)
        if inited is not None and not bool(torch.as_tensor(inited).bool().all()):
            raise EngineError("checkpoint has un-initialised codebooks (quantizer.rq.model.inited == 0); the reference "
                              "would run k-means on the first batch (ddp_core_vq.py:149-159), which is training behaviour")
        for key, shape in want.items():
            if key not in state:
                raise EngineError(f"checkpoint is missing tensor {key} {shape}")
            t = torch.as_tensor(state[key]).detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise EngineError(f"shape mismatch for {key}: checkpoint {tuple(t.shape)} vs architecture {shape}")
            dims = (C.c_int64 * 4)(*t.shape)
            self._check(self.lib.fc_engine_set_weight(self._h, key.encode(), C.c_void_p(t.data_ptr()), dims, t.dim()))
        skipped = [k for k in state if k not in want]
        if skipped:
            logging.info("funcodec_amd: skipped %d checkpoint tensors outside the hot path (e.g. %s)", len(skipped), skipped[0])
        self._check(self.lib.fc_engine_finalize(self._h))
        self._finalized = True

    # -- sizes ---------------------------------------------------------------------------------
    @property
    def hop_length(self) -> int:
        return self.lib.fc_engine_hop_length(self._h)

    def frames(self, n_samples: int) -> int:
        return self.lib.fc_engine_frames(self._h, n_samples)

    def decoded_samples(self, n_frames: int) -> int:
        """Samples the decoder emits for n_frames frames (n_frames * hop; stft_hop * (2-D time frames - 1) for freq_codec)."""
        return self.lib.fc_engine_decoded_samples(self._h, n_frames)

    def _workspace(self, B: int, T: int) -> torch.Tensor:
        key = (B, T)
        need = self._ws_need.get(key)
        if need is None:
            need = self._ws_need[key] = int(self.lib.fc_engine_workspace_bytes(self._h, B, T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def work(self, B: int, T: int, n_q: int) -> Dict[str, float]:
        w = _lib.FcWork()
        self._check(self.lib.fc_engine_work(self._h, B, T, n_q, C.byref(w)))
        return {f: getattr(w, f) for f, _ in _lib.FcWork._fields_}

    def set_profiling(self, on: bool) -> None:
        self._check(self.lib.fc_engine_profile(self._h, int(on)))

    @_on_device
    def read_profile(self):
        """Per-kernel-class totals since the last read (synchronises on the last recorded event)."""
        arr = (_lib.FcProf * _lib.FC_PROF_CLASSES)()
        self._check(self.lib.fc_engine_profile_read(self._h, arr))
        return [dict(kernel=p.kernel.decode(), total_ms=p.total_ms, flops=p.flops, bytes=p.bytes, launches=p.launches)
                for p in arr]

    def _dev(self, t: torch.Tensor, dtype) -> torch.Tensor:
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _wav_in(self, wav: torch.Tensor) -> torch.Tensor:
        """[B,T] (mono engines) or [B,C,T] with C = the model's audio channels, on the device, contiguous fp32."""
        wav = self._dev(wav, torch.float32)
        if wav.dim() == 2 and self.channels == 1:
            return wav
        if wav.dim() != 3 or wav.shape[1] != self.channels:
            raise EngineError(f"wav must be [B,{self.channels},T]" + (" or [B,T]" if self.channels == 1 else "") + f", got {tuple(wav.shape)}")
        return wav

    # -- hot path ------------------------------------------------------------------------------
    @staticmethod
    def _cat(parts, dims):
        """Concatenate per-micro-batch result dicts along each tensor's batch dimension."""
        out = {}
        for k, dim in dims.items():
            vals = [p[k] for p in parts]
            out[k] = None if vals[0] is None else torch.cat(vals, dim)
        return out

    @_on_device
    def encode(self, wav: torch.Tensor, n_q: int, want_sub_quants: bool = True, want_enc_out: bool = False):
        """wav [B,T] (or [B,C,T]) -> dict(codes [n_q,B,Tf] i64, quantized [B,Tf,D], sub_quants [n_q,B,D,Tf], scale [B,1]|None)."""
        wav = self._wav_in(wav)
        B, T = wav.shape[0], wav.shape[-1]
        if B > self.micro_batch:
            parts = [self.encode(wav[i:i + self.micro_batch], n_q, want_sub_quants, want_enc_out)
                     for i in range(0, B, self.micro_batch)]
            return self._cat(parts, dict(codes=1, quantized=0, sub_quants=1, scale=0, enc_out=0))
        Tf, D = self.frames(T), self.arch.dimension
        dev = self.device
        codes = torch.empty((n_q, B, Tf), dtype=torch.int64, device=dev)
        quant = torch.empty((B, Tf, D), dtype=torch.float32, device=dev)
        subq = torch.empty((n_q, B, self.arch.codebook_dim, Tf), dtype=torch.float32, device=dev) if want_sub_quants else None
        scale = torch.empty((B,), dtype=torch.float32, device=dev) if self.arch.audio_normalize else None
        enc = torch.empty((B, Tf, D), dtype=torch.float32, device=dev) if want_enc_out else None
        ws = self._workspace(B, T)
        self._check(self.lib.fc_encode(self._h, _ptr(wav), B, T, n_q, _ptr(codes), _ptr(quant), _ptr(subq), _ptr(scale),
                                       _ptr(enc), _ptr(ws), ws.numel(), self._stream()))
        return dict(codes=codes, quantized=quant, sub_quants=subq,
                    scale=None if scale is None else scale.view(B, 1), enc_out=enc)

    @_on_device
    def encode_decode(self, wav: torch.Tensor, n_q: int, use_scale: bool = True, want_sub_quants: bool = True):
        wav = self._wav_in(wav)
        B, T = wav.shape[0], wav.shape[-1]
        if B > self.micro_batch:
            parts = [self.encode_decode(wav[i:i + self.micro_batch], n_q, use_scale, want_sub_quants)
                     for i in range(0, B, self.micro_batch)]
            return self._cat(parts, dict(codes=1, quantized=0, sub_quants=1, scale=0, recon=0))
        Tf, D = self.frames(T), self.arch.dimension
        dev = self.device
        codes = torch.empty((n_q, B, Tf), dtype=torch.int64, device=dev)
        quant = torch.empty((B, Tf, D), dtype=torch.float32, device=dev)
        subq = torch.empty((n_q, B, self.arch.codebook_dim, Tf), dtype=torch.float32, device=dev) if want_sub_quants else None
        scale = torch.empty((B,), dtype=torch.float32, device=dev) if self.arch.audio_normalize else None
        recon = torch.empty((B, self.channels, min(T, self.decoded_samples(Tf))), dtype=torch.float32, device=dev)   # like recon[:, :, :T] of the reference
        ws = self._workspace(B, T)
        self._check(self.lib.fc_encode_decode(self._h, _ptr(wav), B, T, n_q, int(use_scale), _ptr(codes), _ptr(quant),
                                              _ptr(subq), _ptr(scale), _ptr(recon), _ptr(ws), ws.numel(), self._stream()))
        return dict(codes=codes, quantized=quant, sub_quants=subq,
                    scale=None if scale is None else scale.view(B, 1), recon=recon)

    @_on_device
    def decode_codes(self, tokens: torch.Tensor):
        """tokens [B,Tf,n_q] i64 -> (wav [B,1,Tf*hop], emb [B,Tf,D])."""
        tokens = self._dev(tokens, torch.int64)
        B, Tf, n_q = tokens.shape
        if B > self.micro_batch:
            parts = [self.decode_codes(tokens[i:i + self.micro_batch]) for i in range(0, B, self.micro_batch)]
            return torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)
        L = self.decoded_samples(Tf)
        wav = torch.empty((B, self.channels, L), dtype=torch.float32, device=self.device)
        emb = torch.empty((B, Tf, self.arch.dimension), dtype=torch.float32, device=self.device)
        ws = self._workspace(B, Tf * self.hop_length)
        self._check(self.lib.fc_decode_codes(self._h, _ptr(tokens), B, Tf, n_q, L, _ptr(wav), _ptr(emb), _ptr(ws), ws.numel(),
                                             self._stream()))
        return wav, emb

    @_on_device
    def decode_emb(self, emb: torch.Tensor, scale: Optional[torch.Tensor] = None, out_len: Optional[int] = None):
        """emb [B,Tf,D] -> wav [B,1,out_len or Tf*hop]."""
        emb = self._dev(emb, torch.float32)
        B, Tf, D = emb.shape
        if D != self.arch.dimension:
            raise EngineError(f"embedding dim {D} != {self.arch.dimension}")
        if B > self.micro_batch:
            return torch.cat([self.decode_emb(emb[i:i + self.micro_batch],
                                              None if scale is None else scale.reshape(-1)[i:i + self.micro_batch], out_len)
                              for i in range(0, B, self.micro_batch)], 0)
        L = self.decoded_samples(Tf)
        out_len = L if out_len is None else int(out_len)
        sc = None if scale is None else self._dev(scale.reshape(-1), torch.float32)
        wav = torch.empty((B, self.channels, out_len), dtype=torch.float32, device=self.device)
        ws = self._workspace(B, Tf * self.hop_length)
        self._check(self.lib.fc_decode_emb(self._h, _ptr(emb), _ptr(sc), B, Tf, out_len, _ptr(wav), _ptr(ws), ws.numel(),
                                           self._stream()))
        return wav

    @_on_device
    def overlap_add(self, frames, stride: int, out_len: Optional[int] = None) -> torch.Tensor:
        """_linear_overlap_add (codec_basic.py:77-116) of decoded segments: frames = list of [B,1,L_f] device tensors
        (frame f starts at f*stride) -> [B,1,out_len or total]."""
        shape = tuple(frames[0].shape[:-1]) if frames[0].dim() == 3 else (frames[0].shape[0], 1)      # [B,C]: every channel row is overlap-added alone
        frames = [self._dev(f.reshape(-1, f.shape[-1]), torch.float32) for f in frames]
        B = frames[0].shape[0]
        lens = [int(f.shape[1]) for f in frames]
        total = stride * (len(frames) - 1) + lens[-1]
        out_len = total if out_len is None else min(int(out_len), total)
        ptrs = torch.tensor([f.data_ptr() for f in frames], dtype=torch.int64).to(self.device)     # plumbing: one small H2D
        lens_d = torch.tensor(lens, dtype=torch.int32).to(self.device)
        out = torch.empty((B, 1, out_len), dtype=torch.float32, device=self.device)
        self._check(self.lib.fc_overlap_add(_ptr(ptrs), _ptr(lens_d), len(frames), B, lens[0], int(stride), out_len, _ptr(out),
                                            self._stream()))
        return out.view(*shape, out_len)

    def debug_freq_features(self, buf: Optional[torch.Tensor], mode: int) -> None:
        """Test hook (fc_debug_freq_features): the next encode / encode_decode call of this thread hands its STFT-domain feature tensor
        [B, input_channels, n_fft / 2 + 1, frames] to `buf` (mode 1) or takes it from there (mode 2); mode 0 disarms."""
        if buf is not None and (buf.device != self.device or buf.dtype != torch.float32 or not buf.is_contiguous()):
            raise EngineError("debug_freq_features: a contiguous float32 tensor on the engine's device")
        self._check(self.lib.fc_debug_freq_features(_ptr(buf), 0 if buf is None else buf.numel() * 4, int(mode)))
        # the library keeps a RAW device pointer until the next encode call of this thread returns: keep the tensor alive at least that long
        # (the reference is replaced by the next hook and dropped with the engine)
        self._feat_hook_ref = buf if mode else None

    def check_status(self, sync: bool = True) -> None:
        """Raise if a kernel of an earlier call recorded a failure (persistent-LSTM barrier timeout, out-of-range code
        index).  With sync=True the engine's stream is synchronised first, so every call enqueued so far is covered."""
        if sync and torch.cuda.is_available():
            torch.cuda.current_stream(self.device).synchronize()
        self._check(self.lib.fc_engine_status(self._h, None))

    # -- per-op entry points (tests) -----------------------------------------------------------
    @_on_device
    def rvq_encode(self, x: torch.Tensor, n_q: int):
        x = self._dev(x, torch.float32)
        N, D = x.shape
        if D != self.arch.codebook_dim:
            raise EngineError(f"rvq_encode: rows must have {self.arch.codebook_dim} dims, got {D}")
        codes = torch.empty((n_q, N), dtype=torch.int64, device=self.device)
        quant = torch.empty((N, D), dtype=torch.float32, device=self.device)
        ws = torch.empty(N, dtype=torch.int32, device=self.device) if self.arch.q0_ds_ratio > 1 else None     # stage-0 source-row table
        self._check(self.lib.fc_rvq_encode(self._h, _ptr(x), N, n_q, _ptr(codes), _ptr(quant), _ptr(ws) if ws is not None else None,
                                           0 if ws is None else 4 * N, self._stream()))
        return codes, quant

    @_on_device
    def layer_forward(self, prefix: str, x: torch.Tensor, apply_elu: bool = False) -> torch.Tensor:
        x = self._dev(x, torch.float32)
        B, Cin, T = x.shape
        Tout = self.lib.fc_layer_out_len(self._h, prefix.encode(), T)
        if Tout < 0:
            raise EngineError(f"unknown layer {prefix}")
        inner = ".convtr.bias" if prefix.endswith("convtr") else ".conv.bias"
        cout = self.expected_tensors()[prefix + inner][0]
        y = torch.empty((B, cout, Tout), dtype=torch.float32, device=self.device)
        ws = self._workspace(B, max(T, Tout) * 4 + 4096)
        need = (B * cout * (Tout + 64) * 4) * 2 + (1 << 20)
        if ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            ws = self._ws
        self._check(self.lib.fc_layer_forward(self._h, prefix.encode(), _ptr(x), B, T, int(apply_elu), _ptr(y), _ptr(ws),
                                              ws.numel(), self._stream()))
        return y

    @_on_device
    def resblock_forward(self, prefix: str, x: torch.Tensor) -> torch.Tensor:
        """SEANetResnetBlock.forward of the block at Sequential prefix `prefix` (e.g. "encoder.model.1"): [B,C,T] -> [B,C,T]."""
        x = self._dev(x, torch.float32)
        B, Cc, T = x.shape
        y = torch.empty_like(x)
        need = B * Cc * (T + 64) * 4 * 4 + (4 << 20)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ws = self._ws
        self._check(self.lib.fc_resblock_forward(self._h, prefix.encode(), _ptr(x), B, T, _ptr(y), _ptr(ws), ws.numel(), self._stream()))
        return y

    @_on_device
    def lstm_forward(self, prefix: str, x: torch.Tensor) -> torch.Tensor:
        x = self._dev(x, torch.float32)
        B, H, T = x.shape
        want = self.expected_tensors().get(prefix + ".weight_hh_l0")
        if want is None or want[1] != H:
            raise EngineError(f"lstm {prefix!r}: expected input [B, {want[1] if want else '?'}, T], got {tuple(x.shape)}")
        y = torch.empty_like(x)
        need = (T * B * 4 * H + 3 * B * H + B * H * T + (2 * T + 1) * 16 * ((B + 15) // 16) * H) * 4 * self.arch.lstm_layers + (1 << 20)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ws = self._ws
        self._check(self.lib.fc_lstm_forward(self._h, prefix.encode(), _ptr(x), B, T, _ptr(y), _ptr(ws), ws.numel(), self._stream()))
        return y
