"""Shared helpers for the parity tests (the oracle is imported here and ONLY by tests/bench/smoke)."""
import json
import os
import functools

import numpy as np
import torch

from funcodec_amd.config import arch_from_config, recipe_config
from funcodec_amd.synth import make_state_dict, synthetic_audio

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    with open(os.path.join(GOLD, "MANIFEST.json")) as f:
        return json.load(f)


def golden(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@functools.lru_cache(maxsize=8)
def state_for(cfg_name, seed, decay=1.0):
    cfg = recipe_config(cfg_name)
    arch = arch_from_config(cfg)
    return cfg, arch, make_state_dict(arch, seed, decay)


@functools.lru_cache(maxsize=8)
def engine_for(cfg_name, seed, decay=1.0):
    from funcodec_amd.model import EncodecMI355X
    cfg, arch, sd = state_for(cfg_name, seed, decay)
    m = EncodecMI355X(arch, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m


@functools.lru_cache(maxsize=8)
def oracle_for(cfg_name, seed, decay=1.0):
    from torch_oracle import Oracle
    cfg, arch, sd = state_for(cfg_name, seed, decay)
    return Oracle(cfg, {k: torch.from_numpy(v) for k, v in sd.items()})


@functools.lru_cache(maxsize=4)
def freq_state_for(cfg_name, seed):
    """FreqCodec cases: recipe config, ArchSpec and the seeded 2-D checkpoint (funcodec_amd/synth.py)."""
    from funcodec_amd.config import freq_recipe_config; from funcodec_amd.synth import make_freq_state_dict
    cfg = freq_recipe_config(cfg_name)
    return cfg, arch_from_config(cfg), make_freq_state_dict(cfg, seed)


@functools.lru_cache(maxsize=4)
def freq_engine_for(cfg_name, seed):
    from funcodec_amd.model import EncodecMI355X
    cfg, arch, sd = freq_state_for(cfg_name, seed)
    m = EncodecMI355X(arch, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m


@functools.lru_cache(maxsize=4)
def freq_oracle_for(cfg_name, seed):
    from freq_oracle import FreqOracle
    cfg, arch, sd = freq_state_for(cfg_name, seed)
    return FreqOracle(cfg, {k: torch.from_numpy(v) for k, v in sd.items()})


def audio(B, T, seed, kind="noise", channels=1):
    """Test audio of a golden case: seeded synthetic audio, or (kind "wav:<name>") one of the reference's own demo
    recordings committed under tests/golden/wav/ (decoded like the product's reader: PCM16 / 2^15).  channels = 2 (stereo cases):
    [B, 2, T], channel c of utterance b = row 2 b + c of the mono generator (as oracle/make_golden.py builds them)."""
    if channels > 1:
        return torch.from_numpy(synthetic_audio(B * channels, T, seed, kind)).reshape(B, channels, T)
    if kind.startswith("wav:"):
        from funcodec_amd.io import read_wav
        x, sr = read_wav(os.path.join(GOLD, "wav", kind[4:] + ".wav"))
        assert sr == 16000 and x.shape[0] == T and B == 1
        return torch.from_numpy(x[None].copy())
    return torch.from_numpy(synthetic_audio(B, T, seed, kind))


def rms(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float(((a - b) ** 2).mean().sqrt())


def index_report(got, ref):
    """Fraction of frames whose code stack is identical, and per-stage first-divergence histogram."""
    got = torch.as_tensor(got).cpu().long()
    ref = torch.as_tensor(ref).cpu().long()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    nq = got.shape[0]
    g = got.reshape(nq, -1)
    r = ref.reshape(nq, -1)
    neq = (g != r)
    frames_bad = neq.any(0)
    first = torch.where(frames_bad, neq.float().argmax(0), torch.full_like(neq[0], -1, dtype=torch.long))
    return dict(frames=int(g.shape[1]), frames_bad=int(frames_bad.sum()),
                mismatched_indices=int(neq.sum()), total_indices=int(neq.numel()),
                first_stage=[int(x) for x in first[frames_bad].tolist()])
