"""CPU tests: the oracle itself against the golden vectors produced by the REAL reference
(oracle/make_golden.py), and the plain-C RVQ restatement against the same."""
import numpy as np
import pytest
import torch

from helpers import audio, golden, index_report, manifest, oracle_for, rms

MAN = manifest()
E2E = [n for n, c in MAN["cases"].items() if c.get("kind") not in ("rvq", "rvq_noddp", "segmented", "freq", "freqseg", "variants", "bypass")]
SEG = [n for n, c in MAN["cases"].items() if c.get("kind") == "segmented"]
SAME_BUILD = torch.__version__ == MAN["torch"]


@pytest.mark.parametrize("name", E2E)
def test_torch_oracle_matches_reference_golden(name):
    c = MAN["cases"][name]
    if c["config"] == "ds640" and c["samples"] > 12000 and not SAME_BUILD:
        pytest.skip("different torch build")
    if c["samples"] * c["batch"] > 200000:
        pytest.skip("10 s x 2 on the CPU oracle: covered by the GPU suite and by oracle/make_golden.py's own assertions")
    orc = oracle_for(c["config"], c["weight_seed"], c["codebook_decay"])
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"], c.get("channels", 1))
    g = golden(name)
    o = orc.inference(wav, bit_width=c["bit_width"], use_scale=True)
    rep = index_report(o["code_indices"][0], g["indices"].astype(np.int64))
    # bit-exact on the torch build / thread count that generated the fixtures; ~1e-6 noise otherwise
    if "encoder_out" in g:
        assert rms(o["encoder_out"], g["encoder_out"]) < 1e-5
    assert rms(o["recon_speech"], g["recon"]) < 1e-4
    if SAME_BUILD and torch.get_num_threads() == MAN["threads"]:
        assert rep["mismatched_indices"] == 0
        assert np.array_equal(o["recon_speech"].numpy(), g["recon"])
    else:
        assert rep["frames_bad"] <= max(1, rep["frames"] // 100)
    if "recon_from_codes" in g:
        tok = torch.from_numpy(g["indices"].astype(np.int64)).permute(1, 2, 0).contiguous()
        wav2, _ = orc.decode_codes(tok)
        assert rms(wav2, g["recon_from_codes"]) < 1e-5


def test_c_oracle_rvq_matches_the_use_ddp_false_reference_quantiser():
    """core_vq.ResidualVectorQuantization (`use_ddp: false`, core_vq.py:324-396) run directly in the build container: same
    distance arithmetic as the ddp class (core_vq.py:183-191), so the same plain-C restatement must reproduce it."""
    import c_oracle
    c = MAN["cases"]["rvq_noddp"]
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    embed = rng.standard_normal((c["n_q"], 1024, 128)).astype(np.float32)
    z = rng.standard_normal((4, 125, 128)).astype(np.float32) * 1.5
    g = golden("rvq_noddp")
    rows = slice(100, 300)
    codes, quant = c_oracle.rvq_encode(z.reshape(-1, 128)[rows], embed, c["n_q"])
    assert np.array_equal(codes, g["indices"].astype(np.int64).reshape(c["n_q"], -1)[:, rows])
    assert np.array_equal(quant, g["quantized"].reshape(-1, 128)[rows])


@pytest.mark.parametrize("name", [n for n, c in MAN["cases"].items() if c.get("kind") == "freq"])
def test_freq_oracle_matches_reference_golden(name):
    """FreqCodec (STFT-domain 2-D SEANet): the restatement oracle/freq_oracle.py against the real reference's
    outputs (the engine is held to the same fixtures in tests/test_gpu_parity.py)."""
    from freq_oracle import FreqOracle
    from funcodec_amd.config import freq_recipe_config; from funcodec_amd.synth import make_freq_state_dict
    c = MAN["cases"][name]
    cfg = freq_recipe_config(c["config"])
    orc = FreqOracle(cfg, {k: torch.from_numpy(v) for k, v in make_freq_state_dict(cfg, c["weight_seed"]).items()})
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"], c.get("channels", 1))
    g = golden(name)
    o = orc.inference(wav, None, True)
    angle = c["config"].endswith("ang")                  # codec_domain mag_angle: 2 channels, and the fixture carries the reference's features
    assert o["features"].shape[1:3] == (2 if angle else 3, orc.n_fft // 2 + 1)
    if angle:
        assert np.array_equal(o["features"].numpy(), g["features"]) or not SAME_BUILD
    assert rms(o["encoder_out"], g["encoder_out"]) < 1e-5 and rms(o["recon_speech"], g["recon"]) < 1e-4
    rep = index_report(o["code_indices"][0], g["indices"].astype(np.int64))
    if SAME_BUILD and torch.get_num_threads() == MAN["threads"]:
        assert rep["mismatched_indices"] == 0 and np.array_equal(o["recon_speech"].numpy(), g["recon"])
    else:
        assert rep["frames_bad"] <= max(1, rep["frames"] // 50)
    from funcodec_amd.config import arch_from_config
    arch = arch_from_config(cfg)
    ds640 = c["config"].endswith("640")
    if not c["config"].startswith("freqfuzz"):
        assert (arch.model_type, arch.ratios, arch.ratios_f, arch.hop_length) == \
            ("freq_codec", (2, 1, 2, 1) if ds640 else (1, 1, 2, 1), (4, 4, 4, 4), 640 if ds640 else 320)
    assert arch.frames_for(c["samples"]) == g["indices"].shape[2]
    for key, bad in (("encoder", "encodec_seanet_encoder"), ("model_conf", dict(cfg["model_conf"], codec_domain=["stft", "stft"])),
                     ("model_conf", dict(cfg["model_conf"], codec_domain=["mag_phase", "mag_angle"])), ("input_size", 1)):
        with pytest.raises(NotImplementedError):
            arch_from_config(dict(cfg, **{key: bad}))      # other FreqCodec flavours are refused, not mis-decoded


@pytest.mark.parametrize("name", [n for n, c in MAN["cases"].items() if c.get("kind") == "freqseg"])
def test_freq_oracle_segmented_mode_matches_reference_golden(name):
    """FreqCodec with model_conf.segment_dur set (codec_freq.py:303-328,390-404): per-frame codec + triangle overlap-add."""
    from freq_oracle import FreqOracle
    from funcodec_amd.config import freq_recipe_config; from funcodec_amd.synth import make_freq_state_dict
    c = MAN["cases"][name]
    cfg = freq_recipe_config(c["config"])
    orc = FreqOracle(cfg, {k: torch.from_numpy(v) for k, v in make_freq_state_dict(cfg, c["weight_seed"]).items()})
    g = golden(name)
    o = orc.inference(audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"], c.get("channels", 1)), None, True)
    assert [int(i.shape[2]) for i in o["code_indices"]] == c["frames"]
    assert rms(o["recon_speech"], g["recon"]) < 1e-4
    exact = SAME_BUILD and torch.get_num_threads() == MAN["threads"]
    for f, idx in enumerate(o["code_indices"]):
        rep = index_report(idx, g[f"indices_{f}"].astype(np.int64))
        assert rep["mismatched_indices"] == 0 if exact else rep["frames_bad"] <= 1
    from funcodec_amd.config import arch_from_config
    a = arch_from_config(cfg)
    assert a.segment_length == 2400 and a.segment_stride == 2160


@pytest.mark.parametrize("name", SEG)
def test_torch_oracle_segmented_mode_matches_reference_golden(name):
    """model_conf.segment_dur set: per-frame encode / RVQ / decode and the triangle overlap-add (codec_basic.py:334-396)."""
    c = MAN["cases"][name]
    orc = oracle_for(c["config"], c["weight_seed"], c["codebook_decay"])
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"], c.get("channels", 1))
    g = golden(name)
    o = orc.inference(wav, bit_width=c["bit_width"], use_scale=True)
    assert [int(i.shape[2]) for i in o["code_indices"]] == c["frames"]
    assert rms(o["recon_speech"], g["recon"]) < 1e-4
    exact = SAME_BUILD and torch.get_num_threads() == MAN["threads"]
    for f, idx in enumerate(o["code_indices"]):
        rep = index_report(idx, g[f"indices_{f}"].astype(np.int64))
        assert rep["mismatched_indices"] == 0 if exact else rep["frames_bad"] <= max(1, rep["frames"] // 50)
        assert np.allclose(o["code_embeddings"][f][1].numpy(), g[f"scale_{f}"], rtol=1e-6)
    if exact:
        assert np.array_equal(o["recon_speech"].numpy(), g["recon"])


@pytest.mark.parametrize("name", [n for n, c in MAN["cases"].items() if c.get("kind") == "bypass"])
def test_torch_oracle_bypass_quantizer_matches_reference_golden(name):
    """model_conf.bypass_quantizer (codec_basic.py:700-705): encoder output as code embeddings, zero indices [B, Tf], decode from it."""
    c = MAN["cases"][name]
    orc = oracle_for(c["config"], c["weight_seed"])
    g = golden(name)
    o = orc.inference(audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"]), bit_width=None, use_scale=True)
    for f, Tf in enumerate(c["frames"]):
        assert o["code_indices"][f].shape == (c["batch"], Tf) and int(o["code_indices"][f].abs().max()) == 0
        assert rms(o["code_embeddings"][f][0], g[f"emb_{f}"]) < 1e-5
    assert rms(o["recon_speech"], g["recon"]) < 1e-4


@pytest.mark.parametrize("name", ["rvq_flat", "rvq_decay08"])
def test_c_oracle_rvq_matches_reference_golden(name):
    import c_oracle
    c = MAN["cases"][name]
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    sig = (c["codebook_decay"] ** np.arange(32, dtype=np.float64)).astype(np.float32)[:, None, None]
    embed = rng.standard_normal((32, 1024, 128)).astype(np.float32) * sig
    z = rng.standard_normal((8, 250, 128)).astype(np.float32) * 1.5
    rows = slice(700, 900)     # 200 rows of the 2000 keep the scalar C loop to ~2 s
    codes, quant = c_oracle.rvq_encode(z.reshape(-1, 128)[rows], embed, 32)
    g = golden(name)
    ref = g["indices"].astype(np.int64).reshape(32, -1)[:, rows]
    assert np.array_equal(codes, ref), f"{int((codes != ref).sum())} of {codes.size} indices differ"
    assert np.array_equal(quant, g["quantized"].reshape(-1, 128)[rows])
    # decode: sum of code vectors in stage order
    emb = c_oracle.rvq_decode(codes.T.copy(), embed)
    assert np.array_equal(emb, quant)


@pytest.mark.parametrize("name", ["tinyq0_b3_t1003", "tinyq0_b2_t1013", "ds320q0_b2_t16200_bw4000"])
def test_c_oracle_q0_ds_ratio_matches_reference_golden(name):
    """quantizer_conf.q0_ds_ratio > 1 (ddp_core_vq.py:396-404): the plain-C quantiser with its own restatement of torch's nearest-
    neighbour index arithmetic reproduces the real reference's indices and quantised vectors from the reference's encoder output
    (even and odd frame counts; q0_ds_ratio 3 behaves like 2)."""
    import c_oracle
    from helpers import state_for
    c = MAN["cases"][name]
    g = golden(name)
    cfg, arch, sd = state_for(c["config"], c["weight_seed"], c["codebook_decay"])
    assert arch.q0_ds_ratio > 1
    enc = g["encoder_out"]                                  # [B, Tf, D]
    B, Tf, D = enc.shape
    codes, quant = c_oracle.rvq_encode_q0(enc.reshape(-1, D), sd["quantizer.rq.model.embed"], c["n_q"], Tf)
    assert np.array_equal(codes.reshape(c["n_q"], B, Tf), g["indices"].astype(np.int64))
    assert np.array_equal(quant.reshape(B, Tf, D), g["quantized"])
    # stage 0 really is at half rate: frames 2j and 2j + 1 of an even-length utterance share their first index
    if Tf % 2 == 0:
        i0 = g["indices"][0]
        assert np.array_equal(i0[:, 0::2], i0[:, 1::2])


def test_q0_source_frames_restate_torch_nearest_interpolate():
    """The row table the kernels use (C ABI fc_q0_source_frames, host-only) and the C oracle's own version against torch itself:
    F.interpolate(F.interpolate(arange(Tf), size=[Tf // 2]), size=[Tf]) for every Tf in 2 .. 2100 and a few large ones."""
    import ctypes as C
    import c_oracle
    import torch.nn.functional as F
    from funcodec_amd import _lib
    lib = _lib.load()
    for Tf in list(range(2, 2101)) + [4095, 4096, 4097, 10001, 65535]:
        a = torch.arange(Tf, dtype=torch.float32)[None, None]
        want = F.interpolate(F.interpolate(a, size=[Tf // 2]), size=[Tf])[0, 0].long().numpy()
        out = (C.c_int32 * Tf)()
        assert lib.fc_q0_source_frames(Tf, out) == 0
        assert np.array_equal(np.frombuffer(out, np.int32), want), Tf
        assert np.array_equal(c_oracle.q0_source(Tf), want), Tf
    assert lib.fc_q0_source_frames(1, (C.c_int32 * 1)()) != 0


def test_c_oracle_small_dims_and_ties():
    """D=16/K=64 (the tiny config) and a planted exact tie: the first index must win."""
    import c_oracle
    rng = np.random.Generator(np.random.PCG64(5))
    cb = rng.standard_normal((3, 64, 16)).astype(np.float32)
    cb[0, 40] = cb[0, 7]           # duplicate code vector -> exact tie whenever 7 is the nearest
    x = cb[0, 7][None, :] + 0.01 * rng.standard_normal((5, 16)).astype(np.float32)
    codes, quant = c_oracle.rvq_encode(x, cb, 3)
    assert (codes[0] == 7).all()
    orc_t = torch.from_numpy(cb)
    # torch restatement agrees on this easy case
    from torch_oracle import Oracle
    from funcodec_amd.config import recipe_config
    o = Oracle(recipe_config("tiny"), {"quantizer.rq.model.embed": orc_t})
    _, idx, _ = o.rvq_forward(torch.from_numpy(x)[None], 3)
    assert np.array_equal(idx[:, 0].numpy(), codes)
