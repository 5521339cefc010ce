"""Batch I/O and wire formats around the hot path: what
`funcodec/bin/codec_inference.py:227-378` reads and writes, without torchaudio / kaldiio / librosa.

* wav in:  first channel as float32 in [-1, 1)  (``torchaudio.load(x)[0][0]``, iterable_dataset.py:84)
* scp in:  ``<uttid> <path>`` lines; ``codec_json`` lines ``<uttid> [[[...] x n_q]]`` (iterable_dataset.py:54-58)
* batches: shorter items are WRAP-padded with ``np.pad(mode="wrap")`` (nets_utils.py:65-98, codec_inference.py:257-261)
* out:     ``codecs.txt`` jsonl (codec_inference.py:288-299), Kaldi ``ark,scp`` float matrices (:292-294, :301-311),
           PCM16 wav with peak rescale to 0.99 (``save_audio`` :153-161)
"""
from __future__ import annotations

import json
import os
import struct
import wave
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import ctypes as C

import numpy as np
import torch

_NATIVE = False


def _native():
    """The engine library's host-side wire-format helpers (fc_format_codec_json, fc_write_wav_pcm16), or None when the library has not
    been built (pure-Python formatting then; the compute path has no such fallback)."""
    global _NATIVE
    if _NATIVE is False:
        try:
            from . import _lib
            _NATIVE = _lib.load()
        except Exception:                      # noqa: BLE001 -- I/O helpers only
            _NATIVE = None
    return _NATIVE


# ------------------------------------------------------------------------------------------------
# wav
# ------------------------------------------------------------------------------------------------
def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """First channel of a RIFF wav as float32 (PCM16/32 scaled by 2^15 / 2^31, float passed through)."""
    try:                                   # plain PCM: the stdlib reader is ~25x faster than scipy's chunk walker (0.13 vs 3.5 ms per 10 s)
        with wave.open(path, "rb") as f:
            ch, width, sr, n = f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()
            if width in (2, 4):
                data = np.frombuffer(f.readframes(n), dtype=np.int16 if width == 2 else np.int32).reshape(-1, ch)[:, 0]
                scale = 32768.0 if width == 2 else 2147483648.0
                x = data.astype(np.float32) / np.float32(scale) if width == 2 else (data.astype(np.float64) / scale).astype(np.float32)
                return np.ascontiguousarray(x), int(sr)
    except (wave.Error, EOFError):         # float / extensible / 8-bit files: scipy
        pass
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.ndim > 1:
        data = data[:, 0]
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    return np.ascontiguousarray(x), int(sr)


def save_audio(wav: torch.Tensor, path: str, sample_rate: int, rescale: bool = False) -> None:
    """codec_inference.py:153-161: peak-rescale to 0.99 (or clamp), then 16-bit PCM.  Mono float32 goes through the library's
    fc_write_wav_pcm16 (same arithmetic, one pass in C; byte-identical files, tests/test_io.py)."""
    limit = 0.99
    wav = torch.as_tensor(wav).detach().float().cpu()
    if wav.dim() == 1:
        wav = wav[None]
    if wav.shape[0] == 1 and _native() is not None:
        x = wav[0].contiguous()
        if _native().fc_write_wav_pcm16(os.fsencode(path), x.data_ptr(), x.numel(), int(sample_rate), int(bool(rescale))) != 0:
            raise OSError(_native().fc_last_error().decode("utf-8", "replace"))
        return
    mx = wav.abs().max()
    if rescale:
        wav = wav * min(limit / mx, 1) if mx > 0 else wav
    else:
        wav = wav.clamp(-limit, limit)
    pcm = torch.clamp((wav * 32768.0).round(), -32768, 32767).to(torch.int16).numpy()
    with wave.open(path, "wb") as f:
        f.setnchannels(pcm.shape[0])
        f.setsampwidth(2)
        f.setframerate(int(sample_rate))
        f.writeframes(np.ascontiguousarray(pcm.T).tobytes())


# ------------------------------------------------------------------------------------------------
# sample-rate conversion (reference: torchaudio.functional.resample, codec_inference.py:318-322,352-356)
# ------------------------------------------------------------------------------------------------
def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
             rolloff: float = 0.99) -> torch.Tensor:
    """Band-limited sinc interpolation with a Hann window: the algorithm torchaudio publishes as
    ``torchaudio.functional.resample(..., resampling_method="sinc_interp_hann")`` with its default width / roll-off
    (torchaudio is not installed here, so this is a restatement of that algorithm, not a golden-pinned copy:
    polyphase kernel bank [new/gcd, 1, 2*width + orig/gcd], conv1d with stride orig/gcd, output length
    ceil(new * T / orig)).  Runs wherever `waveform` lives (torch ops: plumbing, not the hot path)."""
    import math
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("sample rates must be positive")
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(orig_freq, new_freq)
    of, nf = orig_freq // g, new_freq // g
    base = min(of, nf) * rolloff
    width = math.ceil(lowpass_filter_width * of / base)
    dt = torch.float64
    idx = torch.arange(-width, width + of, dtype=dt)[None, None] / of
    t = torch.arange(0, -nf, -1, dtype=dt)[:, None, None] / nf + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / of)
    kern = kern.to(dtype=waveform.dtype, device=waveform.device)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1])
    n, length = x.shape
    x = torch.nn.functional.pad(x, (width, width + of))
    y = torch.nn.functional.conv1d(x[:, None], kern, stride=of).transpose(1, 2).reshape(n, -1)
    target = int(math.ceil(nf * length / of))
    return y[..., :target].reshape(shape[:-1] + (target,))


# ------------------------------------------------------------------------------------------------
# Kaldi ark / scp (binary float matrices only: what the reference writes with kaldiio "ark,scp,f:")
# ------------------------------------------------------------------------------------------------
class KaldiMatrixWriter:
    """``kaldiio.WriteHelper("ark,scp,f:<prefix>.ark,<prefix>.scp")`` for float32 matrices."""

    def __init__(self, prefix: str):
        self.ark_path = prefix + ".ark"
        self._ark = open(self.ark_path, "wb")
        self._scp = open(prefix + ".scp", "wt")

    def __call__(self, key: str, mat: np.ndarray) -> None:
        mat = np.ascontiguousarray(mat, dtype=np.float32)
        assert mat.ndim == 2
        self._ark.write(key.encode() + b" ")
        off = self._ark.tell()
        self._ark.write(b"\0BFM " + b"\4" + struct.pack("<i", mat.shape[0]) + b"\4" + struct.pack("<i", mat.shape[1]))
        self._ark.write(mat.tobytes())
        self._scp.write(f"{key} {self.ark_path}:{off}\n")

    def close(self) -> None:
        self._ark.close()
        self._scp.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def load_kaldi_mat(spec: str) -> np.ndarray:
    """``<ark path>:<offset>`` -> float32 matrix (kaldiio.load_mat for the binary FM / DM case)."""
    path, _, off = spec.rpartition(":")
    with open(path, "rb") as f:
        f.seek(int(off))
        if f.read(2) != b"\0B":
            raise ValueError(f"{spec}: not a binary Kaldi object")
        tok = f.read(3)
        if tok not in (b"FM ", b"DM "):
            raise ValueError(f"{spec}: unsupported Kaldi type {tok!r}")
        assert f.read(1) == b"\4"
        rows = struct.unpack("<i", f.read(4))[0]
        assert f.read(1) == b"\4"
        cols = struct.unpack("<i", f.read(4))[0]
        dt = np.float32 if tok == b"FM " else np.float64
        data = np.frombuffer(f.read(rows * cols * np.dtype(dt).itemsize), dtype=dt)
    return data.reshape(rows, cols).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# codec index text format
# ------------------------------------------------------------------------------------------------
def format_codec_line(key: str, indices: Sequence[torch.Tensor], batch_id: int, length: int) -> str:
    """codec_inference.py:295-299: ``<key> [[[T ints] x n_q]]`` (n_frame x n_q x T, n_frame always 1).  One frame of int64 codes on
    the host goes through the library's fc_format_codec_json (byte-identical to json.dumps, ~50x faster; tests/test_io.py)."""
    if len(indices) == 1 and _native() is not None:
        x = indices[0]
        if isinstance(x, torch.Tensor) and x.dtype == torch.int64 and x.device.type == "cpu" and x.dim() == 3 and x.is_contiguous():
            lib = _native()
            n_q, B, T = x.shape
            cap = lib.fc_codec_json_bound(n_q, int(length))
            buf = C.create_string_buffer(cap)
            written = C.c_size_t(0)
            if lib.fc_format_codec_json(x.data_ptr(), n_q, B, T, int(batch_id), int(length), buf, cap, C.byref(written)) == 0:
                return key + " " + buf.raw[:written.value].decode("ascii") + "\n"
    to_write = [x[:, batch_id, :length].cpu().numpy().tolist() for x in indices]
    return key + " " + json.dumps(to_write) + "\n"


def load_codec_json(json_str: str) -> np.ndarray:
    """iterable_dataset.py:54-58 -> [T, n_q] int64."""
    array = np.array(json.loads(json_str))
    if array.ndim == 3:
        array = array[0]
    return array.T


# ------------------------------------------------------------------------------------------------
# scp-driven batching with the reference's wrap padding
# ------------------------------------------------------------------------------------------------
def read_scp(path: str) -> List[Tuple[str, str]]:
    out = []
    with open(path, "rt", encoding="utf-8") as f:
        for line in f:
            line = line.rstrip("\n")
            if not line.strip():
                continue
            key, _, val = line.partition(" ")
            out.append((key, val.strip()))
    return out


def _load_item(value: str, dtype: str) -> np.ndarray:
    if dtype == "sound":
        return read_wav(value)[0]
    if dtype == "codec_json":
        return load_codec_json(value).astype(np.int64)
    if dtype == "kaldi_ark":
        return load_kaldi_mat(value)
    if dtype == "npy":
        return np.load(value)
    raise NotImplementedError(f"data type {dtype!r} is not supported by funcodec_amd.io")


def pad_list_with_mod(xs: Sequence[np.ndarray], pad_value=0.0, mode: str = "wrap") -> torch.Tensor:
    """nets_utils.py:65-98."""
    max_len = max(x.shape[0] for x in xs)
    if all(x.shape[0] == max_len for x in xs):       # equal lengths: nothing to pad
        return torch.from_numpy(np.stack(xs, 0))
    kw = {"mode": mode}
    if mode == "constant":
        kw["constant_values"] = pad_value
    out = []
    for x in xs:
        pads = [(0, max_len - x.shape[0])] + [(0, 0)] * (x.ndim - 1)
        out.append(torch.from_numpy(np.pad(x, pads if x.ndim > 1 else pads[0], **kw)))
    return torch.stack(out, 0)


def iter_batches(data_path_and_name_and_type: Sequence[Tuple[str, str, str]], batch_size: int,
                 key_file: Optional[str] = None) -> Iterator[Tuple[List[str], Dict[str, torch.Tensor]]]:
    """What build_streaming_iterator + common_collate_fn(pad_mode="wrap") yield at inference
    (codec_inference.py:250-264, collate_fn.py:55-95): ``(keys, {name, name_lengths})`` in scp order."""
    tables = [(name, dtype, dict(read_scp(path)), [k for k, _ in read_scp(path)]) for path, name, dtype in data_path_and_name_and_type]
    keys = [k for k, _ in read_scp(key_file)] if key_file else tables[0][3]
    for i in range(0, len(keys), batch_size):
        bkeys = keys[i:i + batch_size]
        batch: Dict[str, torch.Tensor] = {}
        for name, dtype, table, _ in tables:
            items = [_load_item(table[k], dtype) for k in bkeys]
            batch[name] = pad_list_with_mod(items, 0.0 if items[0].dtype.kind == "f" else 0, "wrap")
            batch[name + "_lengths"] = torch.tensor([it.shape[0] for it in items], dtype=torch.long)
        yield bkeys, batch
