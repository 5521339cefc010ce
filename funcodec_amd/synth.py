"""Deterministic synthetic checkpoints in the reference's ``config.yaml`` + ``model.pth`` format.

No pretrained FunCodec checkpoint can be downloaded here (no network), so parity and the benchmark
run on seeded random weights laid out exactly as ``torch.save(model.state_dict())`` of the reference
writes them (funcodec/train/trainer.py:410; key list in SURVEY.md §8a row a19).  The generator is
pure numpy (``np.random.Generator(PCG64(seed))``) so the very same tensors can be re-created on the
GPU box, where the reference and the 230 MB of weights cannot travel; the golden fixtures under
``tests/golden`` store only (config name, seed) plus the reference's outputs for them.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import numpy as np

from .config import ArchSpec, arch_from_config, recipe_config
from .plan import decoder_plan, encoder_plan


def make_state_dict(arch: ArchSpec, seed: int = 0, codebook_sigma_decay: float = 1.0,
                    with_training_extras: bool = True) -> Dict[str, np.ndarray]:
    """Seeded random weights keyed like the reference's state_dict.

    Conv / LSTM weights use the bound torch's default init would give (uniform(+-1/sqrt(fan_in)));
    GroupNorm gamma/beta are randomised around (1, 0) so that a kernel which forgot the affine part
    cannot pass.  ``codebook_sigma_decay`` < 1 gives depth-decaying codebooks (sigma_i = decay**i),
    the case SURVEY.md §7 found to provoke exact fp32 ties in the nearest-neighbour search.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, np.ndarray] = {}

    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    for op in encoder_plan(arch) + decoder_plan(arch):
        if op.kind in ("conv", "convtr"):
            inner = "conv" if op.kind == "conv" else "convtr"
            wshape = (op.cout, op.cin, op.k) if op.kind == "conv" else (op.cin, op.cout, op.k)
            fan_in = op.cin * op.k if op.kind == "conv" else op.cout * op.k  # torch's fan_in for ConvTranspose
            bound = 1.0 / np.sqrt(fan_in)
            if arch.norm == "weight_norm":
                # a trained checkpoint has g != ||v||: randomise g around the norm so that a loader which ignores g fails
                v = uni(wshape, bound)
                nrm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True))
                sd[f"{op.key}.{inner}.weight_v"] = v
                sd[f"{op.key}.{inner}.weight_g"] = (nrm * (1.0 + 0.2 * rng.standard_normal(nrm.shape))).astype(np.float32)
            else:
                sd[f"{op.key}.{inner}.weight"] = uni(wshape, bound)
            sd[f"{op.key}.{inner}.bias"] = uni((op.cout,), bound)
            if arch.norm == "time_group_norm":
                sd[f"{op.key}.norm.weight"] = (1.0 + 0.1 * rng.standard_normal(op.cout)).astype(np.float32)
                sd[f"{op.key}.norm.bias"] = (0.1 * rng.standard_normal(op.cout)).astype(np.float32)
        else:
            h = op.cin
            bound = 1.0 / np.sqrt(h)
            for l in range(arch.lstm_layers):
                sd[f"{op.key}.weight_ih_l{l}"] = uni((4 * h, h), bound)
                sd[f"{op.key}.weight_hh_l{l}"] = uni((4 * h, h), bound)
                sd[f"{op.key}.bias_ih_l{l}"] = uni((4 * h,), bound)
                sd[f"{op.key}.bias_hh_l{l}"] = uni((4 * h,), bound)
    nq, K, D = arch.num_quantizers, arch.codebook_size, arch.codebook_dim
    if D != arch.dimension:
        for nm, (o, i) in (("input_proj", (D, arch.dimension)), ("output_proj", (arch.dimension, D))):
            sd[f"quantizer.{nm}.weight"] = uni((o, i), 1.0 / np.sqrt(i))
            sd[f"quantizer.{nm}.bias"] = uni((o,), 1.0 / np.sqrt(i))
    sig = (codebook_sigma_decay ** np.arange(nq, dtype=np.float64)).astype(np.float32)[:, None, None]
    embed = rng.standard_normal((nq, K, D)).astype(np.float32) * sig
    pfx = "quantizer.rq.model"
    sd[f"{pfx}.inited"] = np.ones((nq, 1), np.float32)
    sd[f"{pfx}.cluster_size"] = np.ones((nq, K), np.float32)
    sd[f"{pfx}.embed"] = embed
    sd[f"{pfx}.embed_avg"] = embed.copy()
    if with_training_extras:
        # keys a real checkpoint carries and the inference engine must skip
        # (discriminator.* and mel_spec_transforms.*)
        sd["discriminator.discriminators.0.dummy.weight"] = np.zeros((4, 4), np.float32)
    return sd


def synthetic_audio(batch: int, n_samples: int, seed: int = 1234, kind: str = "noise") -> np.ndarray:
    """Deterministic test audio [batch, n_samples] fp32.

    ``noise``: 0.1*N(0,1) (BASELINE.md's benchmark input).  ``tones``: a few decaying sinusoids plus
    low-level noise with a different loudness per utterance, closer to speech dynamics.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "noise":
        return (0.1 * rng.standard_normal((batch, n_samples))).astype(np.float32)
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    out = np.zeros((batch, n_samples), np.float64)
    for b in range(batch):
        for _ in range(4):
            f = rng.uniform(80.0, 3500.0)
            ph = rng.uniform(0, 2 * np.pi)
            am = rng.uniform(0.05, 0.3)
            dec = rng.uniform(0.0, 3.0)
            out[b] += am * np.sin(2 * np.pi * f * t + ph) * np.exp(-dec * t)
        out[b] += 0.01 * rng.standard_normal(n_samples)
        out[b] *= rng.uniform(0.2, 1.0)
    return out.astype(np.float32)


# ---- FreqCodec (2-D SEANet) checkpoints: same idea, Conv2d / ConvTranspose2d weight shapes -----------------------------
def freq_plan(cfg: Dict[str, Any]) -> list:
    """(kind, key prefix, weight shape) of every conv / convtr / lstm of the two 2-D nets, in Sequential order."""
    enc, dec = cfg["encoder_conf"], cfg["decoder_conf"]
    nf, dim = enc.get("n_filters", 32), enc.get("dimension", 128)
    ks, lks, rks = enc.get("kernel_size", 7), enc.get("last_kernel_size", 7), enc.get("residual_kernel_size", 3)
    ratios = [tuple(r) for r in enc["ratios"]]
    nres, compress = enc.get("n_residual_layers", 1), enc.get("compress", 2)
    egr, dgr, trgr = enc.get("conv_group_ratio", -1), dec.get("conv_group_ratio", -1), dec.get("tr_conv_group_ratio", -1)
    grp = lambda n, r: n // 2 // r if r > 0 else 1          # groups of a grouped 2-D conv (seanet_encoder.py:224,234,321)
    ops: list = []
    idx, mult = 0, 1
    ops.append(("conv", f"encoder.model.{idx}.conv", (nf, cfg["input_size"], ks, ks)))
    idx += 1
    for fr, tr in reversed(ratios):
        c = mult * nf
        for _ in range(nres):
            h = c // compress
            ops.append(("conv", f"encoder.model.{idx}.block.1.conv", (h, c // grp(h, egr), rks, rks)))
            ops.append(("conv", f"encoder.model.{idx}.block.3.conv", (c, h // grp(h, egr), 1, 1)))
            ops.append(("conv", f"encoder.model.{idx}.shortcut.conv", (c, c // grp(c, egr), 1, 1)))
            idx += 1
        idx += 1
        ops.append(("conv", f"encoder.model.{idx}.conv", (2 * c, c // grp(c, egr), 2 * fr, 2 * tr)))
        idx += 1
        mult *= 2
    idx += 1                                          # ReshapeModule
    cb = mult * nf
    has_lstm = enc.get("seq_model", "lstm") == "lstm"
    if has_lstm:
        ops.append(("lstm", f"encoder.model.{idx}.lstm", (cb, enc.get("seq_layer_num", 2))))
        idx += 1
    idx += 1                                          # ELU
    ops.append(("conv", f"encoder.model.{idx}.conv", (dim, cb, lks)))
    idx = 0
    ops.append(("conv", f"decoder.model.{idx}.conv", (cb, dim, ks)))
    idx += 1
    if has_lstm:
        ops.append(("lstm", f"decoder.model.{idx}.lstm", (cb, enc.get("seq_layer_num", 2))))
        idx += 1
    idx += 1                                          # ReshapeModule
    for fr, tr in ratios:
        c = mult * nf
        idx += 1
        ops.append(("convtr", f"decoder.model.{idx}.convtr", (c, c // 2 // grp(c, trgr), 2 * fr, 2 * tr)))
        idx += 1
        for _ in range(nres):
            c2 = c // 2
            h = c2 // compress
            ops.append(("conv", f"decoder.model.{idx}.block.1.conv", (h, c2 // grp(h, dgr), rks, rks)))
            ops.append(("conv", f"decoder.model.{idx}.block.3.conv", (c2, h // grp(h, dgr), 1, 1)))
            ops.append(("conv", f"decoder.model.{idx}.shortcut.conv", (c2, c2 // grp(c2, dgr), 1, 1)))
            idx += 1
        mult //= 2
    idx += 1
    ops.append(("conv", f"decoder.model.{idx}.conv", (dec.get("channels", 1), nf, lks, lks)))
    return ops


def make_freq_state_dict(cfg: Dict[str, Any], seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, np.ndarray] = {}

    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    for kind, key, shape in freq_plan(cfg):
        if kind == "lstm":
            h = shape[0]
            b = 1.0 / np.sqrt(h)
            for l in range(shape[1] if len(shape) > 1 else 2):
                sd[f"{key}.weight_ih_l{l}"] = uni((4 * h, h), b)
                sd[f"{key}.weight_hh_l{l}"] = uni((4 * h, h), b)
                sd[f"{key}.bias_ih_l{l}"] = uni((4 * h,), b)
                sd[f"{key}.bias_hh_l{l}"] = uni((4 * h,), b)
            continue
        inner = "conv" if kind == "conv" else "convtr"
        cout = shape[0] if kind == "conv" else shape[0] // 2        # every ConvTranspose2d of the net halves the channels
        fan_in = int(np.prod(shape[1:])) if kind == "conv" else int(shape[1] * np.prod(shape[2:]))
        b = 1.0 / np.sqrt(fan_in)
        if cfg["encoder_conf"].get("norm", "weight_norm") == "weight_norm":
            # torch.nn.utils.weight_norm (dim 0; conv.py:24-25): g has one entry per slice of dim 0 -- out channels of Conv, IN channels of
            # ConvTranspose; a trained checkpoint has g != ||v||, so g is randomised around the norm
            v = uni(shape, b)
            nrm = np.sqrt((v.astype(np.float64) ** 2).reshape(shape[0], -1).sum(axis=1)).reshape((shape[0],) + (1,) * (len(shape) - 1))
            sd[f"{key}.{inner}.weight_v"] = v
            sd[f"{key}.{inner}.weight_g"] = (nrm * (1.0 + 0.2 * rng.standard_normal(nrm.shape))).astype(np.float32)
            sd[f"{key}.{inner}.bias"] = uni((cout,), b)
            continue
        sd[f"{key}.{inner}.weight"] = uni(shape, b)
        sd[f"{key}.{inner}.bias"] = uni((cout,), b)
        sd[f"{key}.norm.weight"] = (1.0 + 0.1 * rng.standard_normal(cout)).astype(np.float32)
        sd[f"{key}.norm.bias"] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    q = cfg["quantizer_conf"]
    dim = cfg["encoder_conf"].get("dimension", 128)
    nq, K, D = q["num_quantizers"], q["codebook_size"], q.get("codec_dim") or dim
    if D != dim:      # CostumeQuantizer.input_proj / output_proj (costume_quantizer.py:27-30); drawn AFTER everything else so that the
        prng = np.random.Generator(np.random.PCG64(seed + 7919))       # checkpoints of the configs without a projection do not change
        for nm, (o, i) in (("input_proj", (D, dim)), ("output_proj", (dim, D))):
            b = 1.0 / np.sqrt(i)
            sd[f"quantizer.{nm}.weight"] = prng.uniform(-b, b, size=(o, i)).astype(np.float32)
            sd[f"quantizer.{nm}.bias"] = prng.uniform(-b, b, size=(o,)).astype(np.float32)
    embed = rng.standard_normal((nq, K, D)).astype(np.float32)
    pfx = "quantizer.rq.model"
    sd[f"{pfx}.inited"] = np.ones((nq, 1), np.float32)
    sd[f"{pfx}.cluster_size"] = np.ones((nq, K), np.float32)
    sd[f"{pfx}.embed"] = embed
    sd[f"{pfx}.embed_avg"] = embed.copy()
    return sd


def write_checkpoint(out_dir: str, config: Dict[str, Any], state: Dict[str, np.ndarray]) -> (str, str):
    """Write ``config.yaml`` + ``model.pth`` exactly like a released FunCodec model directory."""
    import torch
    import yaml
    os.makedirs(out_dir, exist_ok=True)
    cfg_path = os.path.join(out_dir, "config.yaml")
    pth_path = os.path.join(out_dir, "model.pth")
    with open(cfg_path, "wt", encoding="utf-8") as f:
        yaml.safe_dump(config, f)
    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}, pth_path)
    return cfg_path, pth_path


def make_checkpoint(out_dir: str, name: str = "ds640", seed: int = 0,
                    codebook_sigma_decay: float = 1.0) -> (str, str):
    cfg = recipe_config(name)
    if cfg.get("model") == "freq_codec":
        return write_checkpoint(out_dir, cfg, make_freq_state_dict(cfg, seed))
    arch = arch_from_config(cfg)
    return write_checkpoint(out_dir, cfg, make_state_dict(arch, seed, codebook_sigma_decay))


# ---- LauraTTS checkpoints (funcodec/models/audio_generation/laura_model.py): same idea for the text -> codec-token model ---------
def laura_plan(cfg: Dict[str, Any]) -> list:
    """(key, shape) of every tensor the generation path of a LauraGenModel checkpoint holds, in state_dict order
    (Text2AudioGenTask.build_model, funcodec/tasks/text2audio_generation.py:202-247).  The model's private training-time
    quantiser (`quantizer.rq.model.*`, laura_model.py:141-151) is not on the inference path and is left out."""
    from .laura_config import laura_spec_from_config
    s = laura_spec_from_config(cfg)
    plan = []

    def stack(prefix, st):
        d, dk = st.d_model, st.d_model // st.heads
        plan.append((f"{prefix}.embed.0.weight", (d, st.idim)))
        plan.append((f"{prefix}.embed.0.bias", (d,)))
        plan.append((f"{prefix}.embed.1.weight", (d,)))
        plan.append((f"{prefix}.embed.1.bias", (d,)))
        for i in range(st.layers):
            p = f"{prefix}.encoders.{i}"
            plan.append((f"{p}.self_attn.pos_bias_u", (st.heads, dk)))
            plan.append((f"{p}.self_attn.pos_bias_v", (st.heads, dk)))
            for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
                plan.append((f"{p}.self_attn.{n}.weight", (d, d)))
                plan.append((f"{p}.self_attn.{n}.bias", (d,)))
            plan.append((f"{p}.self_attn.linear_pos.weight", (d, d)))
            plan.append((f"{p}.feed_forward.w_1.weight", (st.ff, d)))
            plan.append((f"{p}.feed_forward.w_1.bias", (st.ff,)))
            plan.append((f"{p}.feed_forward.w_2.weight", (d, st.ff)))
            plan.append((f"{p}.feed_forward.w_2.bias", (d,)))
            for n in (st.norm_names if st.norm_names[0] == "norm1" else (st.norm_names[1], st.norm_names[0])):
                plan.append((f"{p}.{n}.weight", (d,)))
                plan.append((f"{p}.{n}.bias", (d,)))
        plan.append((f"{prefix}.after_norm.weight", (d,)))
        plan.append((f"{prefix}.after_norm.bias", (d,)))

    D = s.codebook_dim
    stack("text_encoder", s.text_encoder)
    plan.append(("text_enc_out_layer.weight", (D, s.text_encoder.d_model)))
    plan.append(("text_enc_out_layer.bias", (D,)))
    if s.vocab_size > 0:
        plan.append(("token_embedding.weight", (s.vocab_size, s.input_size)))
    plan.append(("lm_embedding.weight", (2, D)))
    stack("codec_lm.encoder", s.codec_lm)
    plan.append(("codec_lm.decoder.weight", (s.lm_vocab, s.codec_lm.d_model)))
    plan.append(("codec_lm.decoder.bias", (s.lm_vocab,)))
    stack("codec_encoder", s.codec_encoder)
    plan.append(("codec_encoder_out_layer.weight", (D, s.codec_encoder.d_model)))
    plan.append(("codec_encoder_out_layer.bias", (D,)))
    plan.append(("quantizer_codebook.embed", (s.num_quantizers, s.codebook_size, D)))
    return plan


def make_laura_state_dict(cfg: Dict[str, Any], seed: int = 0, eos_bias=None) -> Dict[str, np.ndarray]:
    """Seeded random LauraGenModel weights.  Linear layers: torch's default bound; LayerNorm gamma / beta randomised around
    (1, 0); the relative-position biases and the codebook table at unit scale so that every term of the attention score and both
    summed codebooks matter."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, np.ndarray] = {}
    for key, shape in laura_plan(cfg):
        leaf = key.rsplit(".", 1)[-1]
        if key == "quantizer_codebook.embed":
            v = rng.standard_normal(shape)
        elif key in ("lm_embedding.weight", "token_embedding.weight"):
            v = rng.standard_normal(shape)
        elif leaf in ("pos_bias_u", "pos_bias_v"):
            v = 0.5 * rng.standard_normal(shape)
        elif len(shape) == 1 and (".norm" in key or "after_norm" in key or ".embed.1." in key):
            v = 1.0 + 0.1 * rng.standard_normal(shape) if leaf == "weight" else 0.1 * rng.standard_normal(shape)
        else:
            fan_in = shape[-1] if len(shape) == 2 else [s for k, s in laura_plan_fanin(cfg) if k == key][0]
            b = 1.0 / np.sqrt(fan_in)
            v = rng.uniform(-b, b, size=shape)
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    sd["quantizer_codebook.codec_index_shift"] = (1024.0 * np.arange(32, dtype=np.float32))[None, None, :]
    if eos_bias is not None:     # (group, value): raise that group's <eos> logit so that greedy generation ends inside the loop
        from .laura_config import laura_spec_from_config
        K = laura_spec_from_config(cfg).codebook_size
        sd["codec_lm.decoder.bias"][eos_bias[0] * (K + 1) + K] += np.float32(eos_bias[1])
    return sd


def laura_plan_fanin(cfg: Dict[str, Any]) -> list:
    """fan-in of every bias vector (= the input width of its Linear), for torch's default bias bound"""
    shapes = dict(laura_plan(cfg))
    out = []
    for key, shape in shapes.items():
        if key.endswith(".bias") and key[:-5] + ".weight" in shapes and len(shapes[key[:-5] + ".weight"]) == 2:
            out.append((key, shapes[key[:-5] + ".weight"][1]))
    return out


def synthetic_text(cfg: Dict[str, Any], batch: int, lengths, seed: int = 77):
    """Deterministic text-side inputs: embeddings [batch, max(lengths), input_size] (zero past each length) for an
    embedding-input model, token ids [batch, max(lengths)] (int64, -1 past each length) for a token-list model."""
    from .laura_config import laura_spec_from_config
    s = laura_spec_from_config(cfg)
    rng = np.random.Generator(np.random.PCG64(seed))
    L = int(max(lengths))
    if s.vocab_size > 0:
        ids = rng.integers(2, s.vocab_size, size=(batch, L)).astype(np.int64)
        for b, n in enumerate(lengths):
            ids[b, n:] = -1
        return ids
    x = rng.standard_normal((batch, L, s.input_size)).astype(np.float32)
    for b, n in enumerate(lengths):
        x[b, n:] = 0.0
    return x
