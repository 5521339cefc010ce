"""Layer plan of the SEANet encoder / decoder and the checkpoint keys each layer owns.

The module indices reproduce the positions inside the reference's ``nn.Sequential``
(funcodec/models/encoder/seanet_encoder.py:109-160, funcodec/models/decoder/seanet_decoder.py:111-164)
because those integers are part of the checkpoint format (``encoder.model.{i}...`` keys,
SURVEY.md §8a row a19).  The C++ engine (csrc/engine.cpp) builds the same plan from ``fc_arch``;
``tests/test_host.py`` checks the two agree name-for-name.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Tuple

from .config import ArchSpec


@dataclasses.dataclass
class ConvOp:
    kind: str            # "conv" | "convtr" | "lstm"
    key: str             # state_dict prefix, e.g. "encoder.model.3.conv"
    cin: int
    cout: int
    k: int = 1
    stride: int = 1
    role: str = ""       # "first" | "shortcut" | "block1" | "block3" | "down" | "up" | "last" | "lstm"
    dilation: int = 1
    kf: int = 0          # 2-D layers (freq_codec): kernel / stride along frequency; 0 = a 1-D layer
    sf: int = 1
    groups: int = 1      # grouped 2-D convs (conv_group_ratio > 0)


def encoder_plan(a: ArchSpec) -> List[ConvOp]:
    ops: List[ConvOp] = []
    idx = 0
    mult = 1
    ops.append(ConvOp("conv", f"encoder.model.{idx}.conv", a.input_channels, a.n_filters, a.kernel_size, 1, "first"))
    idx += 1
    for ratio in reversed(a.ratios):
        c = mult * a.n_filters
        hid = c // a.compress
        for j in range(a.n_residual_layers):          # dilations [dilation_base**j, 1] (seanet_encoder.py:127-133)
            p = f"encoder.model.{idx}"
            ops.append(ConvOp("conv", f"{p}.shortcut.conv", c, c, 1, 1, "shortcut"))
            ops.append(ConvOp("conv", f"{p}.block.1.conv", c, hid, a.residual_kernel_size, 1, "block1", a.dilation_base ** j))
            ops.append(ConvOp("conv", f"{p}.block.3.conv", hid, c, 1, 1, "block3"))
            idx += 1      # resblock
        idx += 1          # ELU
        ops.append(ConvOp("conv", f"encoder.model.{idx}.conv", c, 2 * c, 2 * ratio, ratio, "down"))
        idx += 1
        mult *= 2
    c = mult * a.n_filters
    if a.lstm_layers > 0:
        ops.append(ConvOp("lstm", f"encoder.model.{idx}.lstm", c, c, role="lstm"))
        idx += 1
    idx += 1              # ELU
    ops.append(ConvOp("conv", f"encoder.model.{idx}.conv", c, a.dimension, a.last_kernel_size, 1, "last"))
    return ops


def decoder_plan(a: ArchSpec) -> List[ConvOp]:
    ops: List[ConvOp] = []
    idx = 0
    mult = 2 ** len(a.ratios)
    c = mult * a.n_filters
    ops.append(ConvOp("conv", f"decoder.model.{idx}.conv", a.dimension, c, a.kernel_size, 1, "first"))
    idx += 1
    if a.lstm_layers > 0:
        ops.append(ConvOp("lstm", f"decoder.model.{idx}.lstm", c, c, role="lstm"))
        idx += 1
    for ratio in a.ratios:
        c = mult * a.n_filters
        idx += 1          # ELU
        ops.append(ConvOp("convtr", f"decoder.model.{idx}.convtr", c, c // 2, 2 * ratio, ratio, "up"))
        idx += 1
        c2 = c // 2
        hid = c2 // a.compress
        for j in range(a.n_residual_layers):
            p = f"decoder.model.{idx}"
            ops.append(ConvOp("conv", f"{p}.shortcut.conv", c2, c2, 1, 1, "shortcut"))
            ops.append(ConvOp("conv", f"{p}.block.1.conv", c2, hid, a.residual_kernel_size, 1, "block1", a.dilation_base ** j))
            ops.append(ConvOp("conv", f"{p}.block.3.conv", hid, c2, 1, 1, "block3"))
            idx += 1
        mult //= 2
    idx += 1              # ELU
    ops.append(ConvOp("conv", f"decoder.model.{idx}.conv", a.n_filters, a.input_channels, a.last_kernel_size, 1, "last"))
    return ops


def encoder_plan_2d(a: ArchSpec) -> List[ConvOp]:
    """SEANetEncoder2d (seanet_encoder.py:252-363): 2-D convs over [frequency, time] down to one frequency bin, then the
    1-D LSTM and last conv.  `k` / `stride` hold the TIME extent, `kf` / `sf` the frequency extent."""
    ops: List[ConvOp] = []
    idx, mult = 0, 1
    gr = a.enc_conv_group_ratio
    grp = lambda n: n // 2 // gr if gr > 0 else 1
    ops.append(ConvOp("conv", f"encoder.model.{idx}.conv", a.input_channels, a.n_filters, a.kernel_size, 1, "first", kf=a.kernel_size))
    idx += 1
    for fr, tr in zip(reversed(a.ratios_f), reversed(a.ratios)):
        c = mult * a.n_filters
        hid = c // a.compress
        for j in range(a.n_residual_layers):
            p = f"encoder.model.{idx}"
            ops.append(ConvOp("conv", f"{p}.shortcut.conv", c, c, 1, 1, "shortcut", kf=1, groups=grp(c)))
            ops.append(ConvOp("conv", f"{p}.block.1.conv", c, hid, a.residual_kernel_size, 1, "block1", a.dilation_base ** j,
                              kf=a.residual_kernel_size, groups=grp(hid)))
            ops.append(ConvOp("conv", f"{p}.block.3.conv", hid, c, 1, 1, "block3", kf=1, groups=grp(hid)))
            idx += 1
        idx += 1          # ELU
        ops.append(ConvOp("conv", f"encoder.model.{idx}.conv", c, 2 * c, 2 * tr, tr, "down", kf=2 * fr, sf=fr, groups=grp(c)))
        idx += 1
        mult *= 2
    idx += 1              # ReshapeModule
    c = mult * a.n_filters
    if a.lstm_layers > 0:
        ops.append(ConvOp("lstm", f"encoder.model.{idx}.lstm", c, c, role="lstm"))
        idx += 1
    idx += 1              # ELU
    ops.append(ConvOp("conv", f"encoder.model.{idx}.conv", c, a.dimension, a.last_kernel_size, 1, "last"))
    return ops


def decoder_plan_2d(a: ArchSpec) -> List[ConvOp]:
    """SEANetDecoder2d (seanet_decoder.py:244-360)."""
    ops: List[ConvOp] = []
    idx, mult = 0, 2 ** len(a.ratios)
    gr, trgr = a.dec_conv_group_ratio, a.dec_tr_conv_group_ratio
    grp = lambda n: n // 2 // gr if gr > 0 else 1
    c = mult * a.n_filters
    ops.append(ConvOp("conv", f"decoder.model.{idx}.conv", a.dimension, c, a.kernel_size, 1, "first"))
    idx += 1
    if a.lstm_layers > 0:
        ops.append(ConvOp("lstm", f"decoder.model.{idx}.lstm", c, c, role="lstm"))
        idx += 1
    idx += 1              # ReshapeModule
    for fr, tr in zip(a.ratios_f, a.ratios):
        c = mult * a.n_filters
        idx += 1          # ELU
        ops.append(ConvOp("convtr", f"decoder.model.{idx}.convtr", c, c // 2, 2 * tr, tr, "up", kf=2 * fr, sf=fr,
                          groups=c // 2 // trgr if trgr > 0 else 1))
        idx += 1
        c2 = c // 2
        hid = c2 // a.compress
        for j in range(a.n_residual_layers):
            p = f"decoder.model.{idx}"
            ops.append(ConvOp("conv", f"{p}.shortcut.conv", c2, c2, 1, 1, "shortcut", kf=1, groups=grp(c2)))
            ops.append(ConvOp("conv", f"{p}.block.1.conv", c2, hid, a.residual_kernel_size, 1, "block1", a.dilation_base ** j,
                              kf=a.residual_kernel_size, groups=grp(hid)))
            ops.append(ConvOp("conv", f"{p}.block.3.conv", hid, c2, 1, 1, "block3", kf=1, groups=grp(hid)))
            idx += 1
        mult //= 2
    idx += 1              # ELU
    ops.append(ConvOp("conv", f"decoder.model.{idx}.conv", a.n_filters, a.input_channels, a.last_kernel_size, 1, "last", kf=a.last_kernel_size))
    return ops


def expected_tensors(a: ArchSpec) -> Dict[str, Tuple[int, ...]]:
    """Every checkpoint tensor the hot path consumes: key -> shape (reference layout)."""
    out: Dict[str, Tuple[int, ...]] = {}
    gn = a.norm == "time_group_norm"
    wn = a.norm == "weight_norm"      # torch.nn.utils.weight_norm: weight = weight_v * (weight_g / ||weight_v||), norm over dims != 0
    ops = encoder_plan_2d(a) + decoder_plan_2d(a) if a.model_type == "freq_codec" else encoder_plan(a) + decoder_plan(a)
    for op in ops:
        if op.kind in ("conv", "convtr"):
            inner = op.kind
            kk = (op.kf, op.k) if op.kf else (op.k,)          # Conv2d weights are [out, in, k_frequency, k_time]
            # torch layouts: Conv [out, in / groups, ...], ConvTranspose [in, out / groups, ...]
            wshape = (op.cout, op.cin // op.groups) + kk if op.kind == "conv" else (op.cin, op.cout // op.groups) + kk
            if wn:
                out[f"{op.key}.{inner}.weight_g"] = (wshape[0],) + (1,) * (len(wshape) - 1)
                out[f"{op.key}.{inner}.weight_v"] = wshape
            else:
                out[f"{op.key}.{inner}.weight"] = wshape
            out[f"{op.key}.{inner}.bias"] = (op.cout,)
            if gn:
                out[f"{op.key}.norm.weight"] = (op.cout,)
                out[f"{op.key}.norm.bias"] = (op.cout,)
        else:
            h = op.cin
            for l in range(a.lstm_layers):
                out[f"{op.key}.weight_ih_l{l}"] = (4 * h, h)
                out[f"{op.key}.weight_hh_l{l}"] = (4 * h, h)
                out[f"{op.key}.bias_ih_l{l}"] = (4 * h,)
                out[f"{op.key}.bias_hh_l{l}"] = (4 * h,)
    if a.codebook_dim != a.dimension:     # CostumeQuantizer.input_proj / output_proj (costume_quantizer.py:27-30)
        out["quantizer.input_proj.weight"] = (a.codebook_dim, a.dimension)
        out["quantizer.input_proj.bias"] = (a.codebook_dim,)
        out["quantizer.output_proj.weight"] = (a.dimension, a.codebook_dim)
        out["quantizer.output_proj.bias"] = (a.dimension,)
    out["quantizer.rq.model.embed"] = (a.num_quantizers, a.codebook_size, a.codebook_dim)
    return out
