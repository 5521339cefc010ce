"""CPU tests of the host logic and of the C-ABI surface (no compute calls: there is no GPU here)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from funcodec_amd import _lib
from funcodec_amd.config import ArchSpec, arch_from_config, recipe_config
from funcodec_amd.engine import CodecEngine, EngineError
from funcodec_amd.parallel import shard_range
from funcodec_amd.plan import expected_tensors
from funcodec_amd.synth import make_state_dict, synthetic_audio
from helpers import GOLD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "funcodec_amd.h")).read()
    declared = set(re.findall(r"\b(fc_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"fc_engine", "fc_arch", "fc_work", "fc_prof"}
    lib = ctypes.CDLL(_lib.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/funcodec_amd.h but not exported"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert _lib.load().fc_abi_version() == _lib.FC_ABI_VERSION


@pytest.mark.parametrize("name", ["ds320", "ds640"])
def test_checkpoint_contract_matches_reference_keys(name):
    """Key names/shapes the engine asks for == the real reference's state_dict (fixture from make_golden.py)."""
    arch = arch_from_config(recipe_config(name))
    ref = json.load(open(os.path.join(GOLD, f"state_dict_keys_{name}.json")))
    want = expected_tensors(arch)
    eng = CodecEngine(arch)
    assert eng.expected_tensors() == want
    for k, shape in want.items():
        assert k in ref, k
        assert tuple(ref[k]) == tuple(shape), (k, ref[k], shape)
    unused = {k for k in ref if k not in want}
    # everything of encoder./decoder. is consumed; only EMA bookkeeping of the quantiser is not
    assert unused == {"quantizer.rq.model.inited", "quantizer.rq.model.cluster_size", "quantizer.rq.model.embed_avg"}


def test_freq_codec_checkpoint_contract_matches_reference_keys():
    """FreqCodec (2-D SEANet): Conv2d weights [out, in, k_frequency, k_time], same key scheme."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "oracle"))
    from funcodec_amd.config import freq_recipe_config
    arch = arch_from_config(freq_recipe_config("freqmp"))
    ref = json.load(open(os.path.join(GOLD, "state_dict_keys_freqmp.json")))
    want = expected_tensors(arch)
    assert CodecEngine(arch).expected_tensors() == want
    assert want["encoder.model.6.conv.conv.weight"] == (128, 64, 8, 4) and want["decoder.model.4.convtr.convtr.weight"] == (512, 256, 8, 2)
    for k, shape in want.items():
        assert tuple(ref[k]) == tuple(shape), (k, ref.get(k), shape)
    assert {k for k in ref if k not in want} == {"quantizer.rq.model.inited", "quantizer.rq.model.cluster_size", "quantizer.rq.model.embed_avg"}


@pytest.mark.parametrize("name", ["tinyfreqwn", "tinyfreqwnc"])
def test_freq_codec_weight_norm_contract(name):
    """2-D nets with weight_norm (no GroupNorm modules), non-causal and causal: weight_g [d0, 1, 1, 1] / weight_v keys, no norm.* keys;
    the engine, the host plan and the seeded checkpoint (which the real reference loaded for the goldens) agree."""
    from funcodec_amd.synth import make_freq_state_dict
    cfg = recipe_config(name)
    arch = arch_from_config(cfg)
    assert arch.norm == "weight_norm" and arch.causal == name.endswith("c")
    want = expected_tensors(arch)
    assert CodecEngine(arch).expected_tensors() == want
    assert want["encoder.model.0.conv.conv.weight_g"] == (4, 1, 1, 1) and not any(".norm." in k for k in want)
    assert want["decoder.model.4.convtr.convtr.weight_g"][1:] == (1, 1, 1)
    sd = make_freq_state_dict(cfg, 0)
    for k, shape in want.items():
        assert sd[k].shape == shape, (k, sd[k].shape, shape)
    assert set(sd) - set(want) == {"quantizer.rq.model.inited", "quantizer.rq.model.cluster_size", "quantizer.rq.model.embed_avg"}


@pytest.mark.parametrize("name", ["tinyfreqgr1", "freqmpgr8", "freqmp640gr1"])
def test_freq_codec_grouped_conv_contract(name):
    """conv_group_ratio / tr_conv_group_ratio > 0 (the "gr" of the released FreqCodec models): the engine, the host plan and the
    seeded checkpoint (which the real reference loaded for the golden) agree on the grouped weight shapes."""
    from funcodec_amd.synth import make_freq_state_dict
    cfg = recipe_config(name)
    arch = arch_from_config(cfg)
    want = expected_tensors(arch)
    assert CodecEngine(arch).expected_tensors() == want
    sd = make_freq_state_dict(cfg, 0)
    for k, shape in want.items():
        assert sd[k].shape == shape, (k, sd[k].shape, shape)
    if name == "freqmpgr8":                       # groups = channels / 2 / 8
        assert want["encoder.model.3.conv.conv.weight"] == (64, 16, 8, 2) and want["decoder.model.4.convtr.convtr.weight"] == (512, 8, 8, 2)
    bad = recipe_config("tinyfreq")
    bad["encoder_conf"]["conv_group_ratio"] = 8                # 4 // 2 // 8 = 0 groups
    with pytest.raises(NotImplementedError):
        arch_from_config(bad)


def test_arch_from_recipe_configs():
    a = arch_from_config(recipe_config("ds640"))
    assert a.ratios == (8, 5, 4, 2, 2) and a.hop_length == 640 and a.bottleneck_channels == 1024
    assert a.num_quantizers_for_bandwidth(None) == 32
    assert a.num_quantizers_for_bandwidth(4000) == 16       # 250 bps per quantiser (vq.py:114-117)
    assert a.num_quantizers_for_bandwidth(100) == 1
    assert a.num_quantizers_for_bandwidth(10 ** 9) == 32
    b = arch_from_config(recipe_config("ds320"))
    assert b.hop_length == 320 and b.num_quantizers_for_bandwidth(8000) == 16
    assert a.frames_for(160000) == 250 and a.frames_for(160001) == 251 and a.frames_for(1) == 1
    assert a.segment_length is None and a.segment_stride is None
    s = arch_from_config(recipe_config("ds320seg"))       # codec_basic.py:288-298
    assert s.segment_length == 8000 and s.segment_stride == 7200
    w = arch_from_config(recipe_config("ds320wn"))        # weight-normalised causal convs (conv.py:20-56,243-305)
    assert w.norm == "weight_norm" and w.causal and a.norm == "time_group_norm" and not a.causal
    from funcodec_amd.plan import expected_tensors
    keys = expected_tensors(w)
    assert keys["encoder.model.3.conv.conv.weight_g"] == (64, 1, 1) and keys["encoder.model.3.conv.conv.weight_v"] == (64, 32, 4)
    assert keys["decoder.model.3.convtr.convtr.weight_g"] == (512, 1, 1)       # dim 0 of a ConvTranspose1d weight = IN channels
    assert not any(k.endswith((".norm.weight", ".conv.weight", ".convtr.weight")) for k in keys)


@pytest.mark.parametrize("mut", [
    lambda c: c["encoder_conf"].update(norm="weight_norm"),                     # encoder / decoder flavours differ
    lambda c: (c["encoder_conf"].update(causal=True), c["decoder_conf"].update(causal=True)),   # GroupNorm + causal (conv.py:46-47)
    lambda c: (c["encoder_conf"].update(norm="layer_norm"), c["decoder_conf"].update(norm="layer_norm")),
    lambda c: c.update(model="freqcodec"),
    lambda c: c["model_conf"].update(segment_dur=1.0, overlap_ratio=1.5),
    lambda c: c["quantizer_conf"].update(codec_dim=48),                          # a width the quantiser kernels are not built for
    lambda c: c["quantizer_conf"].update(codec_range=-1.0),
    lambda c: c["quantizer_conf"].update(q0_ds_ratio=0),
    lambda c: c["quantizer_conf"].update(q0_ds_ratio=2, use_ddp=False),          # only the distributed quantiser has it (core_vq.py takes no such keyword)
    lambda c: c["decoder_conf"].update(ratios=[8, 5, 4]),
])
def test_out_of_scope_configs_are_refused(mut):
    cfg = recipe_config("ds320")
    mut(cfg)
    with pytest.raises(NotImplementedError):
        arch_from_config(cfg)


def test_config_defaults_follow_the_reference_constructor():
    """An ESPnet-style config.yaml stores only the keys the user set; omitted model_conf keys take Encodec.__init__'s
    defaults (codec_basic.py:132-141): audio_normalize=True, segment_dur=1.0, overlap_ratio=0.01, target_sample_hz=24000.
    An explicit null is kept (the recipes set segment_dur: null)."""
    cfg = recipe_config("ds320")
    cfg["model_conf"] = {"odim": 128}
    a = arch_from_config(cfg)
    assert a.audio_normalize is True and a.segment_dur == 1.0 and a.overlap_ratio == 0.01 and a.sample_rate == 24000
    assert a.segment_length == 24000 and a.segment_stride == 23760
    cfg["model_conf"] = {"segment_dur": None, "audio_normalize": False, "target_sample_hz": 16000}
    b = arch_from_config(cfg)
    assert b.segment_length is None and b.audio_normalize is False and b.sample_rate == 16000
    # encoder / decoder values that differ are refused, not silently taken from the encoder
    for key, val in (("activation_params", {"alpha": 0.5}), ("norm_params", {"eps": 1e-3}), ("seq_model", "none")):
        cfg = recipe_config("ds320")
        cfg["decoder_conf"][key] = val
        with pytest.raises(NotImplementedError):
            arch_from_config(cfg)
    assert arch_from_config(recipe_config("tinyq0")).q0_ds_ratio == 3 and arch_from_config(recipe_config("ds320")).q0_ds_ratio == 1
    assert arch_from_config(recipe_config("tinyfreqq0")).q0_ds_ratio == 2
    a = arch_from_config(recipe_config("ds320cd64"))                    # CostumeQuantizer projection + tanh range
    assert (a.dimension, a.codebook_dim, a.codec_range) == (128, 64, 2.5)
    from funcodec_amd.plan import expected_tensors as _et
    assert _et(a)["quantizer.input_proj.weight"] == (64, 128) and _et(a)["quantizer.rq.model.embed"] == (32, 1024, 64)
    eng = CodecEngine(a)
    assert eng.expected_tensors()["quantizer.output_proj.weight"] == (128, 64)
    for name in ("ss320nc", "tinyssnc", "ss640nc", "ds640seg"):                   # the accepted noncausal SoundStream / segmented recipes
        arch_from_config(recipe_config(name))
    # FreqCodec: decoder keys the engine hard-wires are refused when they differ, unknown keys too, and an STFT hop whose window
    # envelope reaches zero (torch.istft raises there)
    for mut in (lambda c: c["decoder_conf"].update(last_out_padding=[(0, 0), (0, 0)]),
                lambda c: c["decoder_conf"].update(some_future_key=1),
                lambda c: c["encoder_conf"].update(some_future_key=1),
                lambda c: c["model_conf"].update(domain_conf={"n_fft": 512, "hop_length": 512})):
        cfg = recipe_config("freqmp")
        mut(cfg)
        with pytest.raises(NotImplementedError):
            arch_from_config(cfg)
    cfg = recipe_config("freqmp")
    cfg["decoder_conf"]["last_out_padding"] = [[0, 1], [0, 0]]          # the default, spelt as yaml spells it
    arch_from_config(cfg)
    bad = arch_from_config(recipe_config("tiny"))
    bad.codebook_size = 192                                            # > 128 and not a multiple of 128: refused at create time
    with pytest.raises(EngineError, match="codebook_size"):
        CodecEngine(bad)


def test_engine_sizes_and_work_accounting():
    arch = arch_from_config(recipe_config("ds640"))
    eng = CodecEngine(arch)
    assert eng.hop_length == 640
    for T in (1, 639, 640, 641, 16000, 160000, 160001):
        assert eng.frames(T) == arch.frames_for(T)
    w1 = eng.lib.fc_engine_workspace_bytes(eng._h, 1, 16000)
    w2 = eng.lib.fc_engine_workspace_bytes(eng._h, 2, 16000)
    w3 = eng.lib.fc_engine_workspace_bytes(eng._h, 2, 32000)
    assert 0 < w1 < w2 < w3
    work = eng.work(16, 160000, 32)
    # 8.51 GFLOP per audio-second for ds640 -> 1.36 TFLOP for the 160 audio-second batch
    assert abs(work["total_flops"] / 1.36e12 - 1.0) < 0.01


def test_engine_fails_loudly_without_gpu_and_on_bad_checkpoints():
    arch = arch_from_config(recipe_config("tiny"))
    sd = make_state_dict(arch, 1)
    eng = CodecEngine(arch)
    bad = dict(sd)
    del bad["encoder.model.3.conv.conv.weight"]
    with pytest.raises(EngineError, match="missing tensor"):
        eng.load_state_dict(bad)
    bad = dict(sd)
    bad["encoder.model.3.conv.conv.weight"] = bad["encoder.model.3.conv.conv.weight"][:, :, :-1]
    with pytest.raises(EngineError, match="shape mismatch"):
        CodecEngine(arch).load_state_dict(bad)
    bad = dict(sd)
    bad["quantizer.rq.model.inited"] = np.zeros_like(bad["quantizer.rq.model.inited"])
    with pytest.raises(EngineError, match="un-initialised"):
        CodecEngine(arch).load_state_dict(bad)
    if not torch.cuda.is_available():
        with pytest.raises(EngineError, match="no CPU fallback"):
            CodecEngine(arch).load_state_dict(sd)      # finalize() needs a gfx950 device
    with pytest.raises(EngineError, match="no CPU fallback|gfx950"):
        CodecEngine(arch, "cpu")


def test_use_ddp_false_checkpoint_layout_is_accepted_up_to_finalize():
    """core_vq.py:147-150 stores one codebook per layer; the loader stacks them."""
    arch = arch_from_config(recipe_config("tiny"))
    sd = make_state_dict(arch, 2)
    emb = sd.pop("quantizer.rq.model.embed")
    for i in range(emb.shape[0]):
        sd[f"quantizer.rq.model.layers.{i}._codebook.embed"] = emb[i]
    sd.pop("quantizer.rq.model.inited")
    eng = CodecEngine(arch)
    try:
        eng.load_state_dict(sd)
    except EngineError as e:           # only the device step may fail on a GPU-less box
        assert "no CPU fallback" in str(e)


def test_speech2token_refuses_cpu(tmp_path):
    from funcodec_amd.bin.codec_inference import Speech2Token
    from funcodec_amd.synth import make_checkpoint
    cfg, pth = make_checkpoint(str(tmp_path), "tiny", 3)
    with pytest.raises(RuntimeError, match="MI355X only"):
        Speech2Token(cfg, pth, device="cpu")


def test_synth_is_deterministic():
    arch = arch_from_config(recipe_config("tiny"))
    a, b = make_state_dict(arch, 7), make_state_dict(arch, 7)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert np.array_equal(synthetic_audio(2, 100, 5, "tones"), synthetic_audio(2, 100, 5, "tones"))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 16, 1024, 1027):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_fc_arch_layout_is_the_same_in_header_binding_and_integration_doc():
    """ABI drift guard: the field list of struct fc_arch in include/funcodec_amd.h, the ctypes mirror in
    funcodec_amd/_lib.py and the stub shown to reference maintainers in INTEGRATION.md must agree name for name."""
    hdr = open(os.path.join(ROOT, "include", "funcodec_amd.h")).read()
    body = hdr[hdr.index("typedef struct fc_arch {"):hdr.index("} fc_arch;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = re.findall(r"\b(?:int32_t|float)\s+([a-z_0-9]+)\s*(?:\[[A-Z_]+\])?\s*;", body)
    py_fields = [f[0] for f in _lib.FcArch._fields_]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    doc_struct = doc[doc.index("class FcArch(C.Structure)"):doc.index("lib = C.CDLL")]
    doc_fields = re.findall(r'\("([a-z_0-9]+)",\s*C\.', doc_struct)
    assert c_fields == py_fields == doc_fields, (c_fields, py_fields, doc_fields)
    assert f"abi_version={_lib.FC_ABI_VERSION}" in doc and f"#define FC_ABI_VERSION {_lib.FC_ABI_VERSION}" in hdr


@pytest.mark.parametrize("k,stride,dil,CC,BM,BN,row", [
    (3, 1, 1, 32, 128, 128, 1), (3, 1, 9, 16, 128, 128, 0), (7, 1, 1, 32, 64, 256, 1), (1, 1, 1, 64, 128, 128, 1),
    (4, 2, 1, 32, 128, 128, 0), (10, 5, 1, 16, 128, 128, 0), (16, 8, 1, 8, 32, 256, 0), (2, 1, 1, 12, 32, 128, 0),
])
def test_conv_operand_layout_is_self_consistent(k, stride, dil, CC, BM, BN, row):
    """Round-5 quad-k operand layout, host side only: the weight packing (conv_pack_index) and the B-operand offset
    table (conv_koff_table) are written by different functions and met only inside the kernel's matrix loop.  This replays that loop's
    address arithmetic in numpy -- quad q = half-quads 2q (MFMA lanes 0-31) and 2q + 1 (lanes 32-63); A piece = image[((2q + hi) BM + row) 4 + s],
    B piece = slab[table[2q + hi] + 4 column + s] -- on a slab laid out as the staging paths write it ([channel quad][position][4 channels];
    strided convs phase-split: position of slab column p = (p % stride) PL + p // stride) and compares with the convolution itself."""
    lib = _lib.load()
    info = (ctypes.c_int * 6)()
    pack = np.zeros(k * CC * BM, np.int32)
    koff = np.zeros(4096, np.int32)
    rc = lib.fc_debug_conv_layout(k, stride, dil, CC, BM, BN, row, info, pack.ctypes.data, pack.size, koff.ctypes.data, koff.size)
    assert rc == 0
    quad, wbuf, nkoff, row_stride, PL, slabW = list(info)
    assert quad == 1 and slabW == (BN - 1) * stride + (k - 1) * dil + 1
    pack = pack.reshape(k, CC, BM)
    assert len(np.unique(pack)) == pack.size and pack.min() >= 0 and pack.max() < wbuf          # an injection into the chunk image
    rng = np.random.default_rng(k * 1000 + CC)
    W = rng.standard_normal((BM, CC, k))
    x = rng.standard_normal((CC, slabW))
    image = np.zeros(wbuf)
    image[pack.transpose(2, 1, 0).reshape(-1)] = W.reshape(-1)                                   # W[m][cl][kk] -> pack[kk][cl][m]
    p = np.arange(slabW)
    pos = p if row else (p % stride) * PL + p // stride
    slab = np.zeros((CC // 4) * row_stride * 4 + 4 * BN + 64)
    for c in range(CC):
        slab[((c // 4) * row_stride + pos) * 4 + (c & 3)] = x[c]
    n_half = k * (CC // 4)
    nq = (wbuf // (8 * BM))                                                                      # quads the image has room for (pads are zero)
    out = np.zeros((BM, BN))
    cols = np.arange(BN)
    for h in range(min(2 * nq, nkoff)):
        q, hi = divmod(h, 2)
        if h >= n_half:
            assert not image[((2 * q + hi) * BM) * 4:((2 * q + hi + 1) * BM) * 4].any()          # pad half-quads multiply zeros
            continue
        for s in range(4):
            a = image[((2 * q + hi) * BM + np.arange(BM)) * 4 + s]
            b = slab[koff[h] + 4 * cols + s]
            out += np.outer(a, b)
    want = np.zeros((BM, BN))
    for kk in range(k):
        want += W[:, :, kk] @ x[:, cols * stride + kk * dil]
    np.testing.assert_allclose(out, want, rtol=0, atol=1e-9)
