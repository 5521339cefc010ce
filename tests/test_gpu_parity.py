"""`-m gpu` parity tests: every call goes through the C ABI (libfuncodec_amd.so) on a real MI355X and is
compared with (a) the golden vectors produced by the real reference, (b) the oracle on the same seeded inputs,
(c) size-independent properties at the benchmark size."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import record_report, record_waivers
from helpers import (audio, engine_for, golden, index_report, manifest, oracle_for, rms, state_for)

pytestmark = pytest.mark.gpu
MAN = manifest()
E2E = [n for n, c in MAN["cases"].items() if c.get("kind") not in ("rvq", "rvq_noddp", "segmented", "freq", "freqseg", "variants", "bypass")]
SEG = [n for n, c in MAN["cases"].items() if c.get("kind") == "segmented"]
FREQ = [n for n, c in MAN["cases"].items() if c.get("kind") == "freq" and not c["config"].endswith("ang")]
FREQ_ANGLE = [n for n, c in MAN["cases"].items() if c.get("kind") == "freq" and c["config"].endswith("ang")]

# tolerances (north_star): integer codec indices bit-exact; waveforms within 1e-4 RMS
WAV_RMS_TOL = 1e-4
LAYER_ABS_TOL = 5e-5          # single layer vs torch CPU, GroupNorm'd outputs are O(1)


def test_native_library_is_loaded_and_device_is_gfx950():
    m = engine_for("tiny", 7)
    maps = open("/proc/self/maps").read()
    assert "libfuncodec_amd.so" in maps
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


# ---- (a) golden vectors of the real reference -----------------------------------------------------
@pytest.mark.parametrize("name", E2E)
def test_e2e_against_reference_golden(name):
    c = MAN["cases"][name]
    m = engine_for(c["config"], c["weight_seed"], c["codebook_decay"])
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"], c.get("channels", 1))
    g = golden(name)
    r = m.engine.encode(wav, c["n_q"], want_enc_out=True)
    if "encoder_out" in g:                         # the two 10 s fixtures store indices / scale / quantized / recon only
        assert rms(r["enc_out"], g["encoder_out"]) < 2e-5
    if "scale" in g:                               # absent when model_conf.audio_normalize is false
        assert float(((r["scale"].cpu() - torch.from_numpy(g["scale"])).abs() / torch.from_numpy(g["scale"])).max()) < 1e-5
    else:
        assert r.get("scale") is None and not m.arch.audio_normalize
    rep = index_report(r["codes"], g["indices"].astype(np.int64))
    r2 = m.engine.encode_decode(wav, c["n_q"], use_scale=True)
    assert torch.equal(r2["codes"], r["codes"])
    assert r2["recon"].shape == (c["batch"], c.get("channels", 1), c["samples"])
    cfg, arch, sd = state_for(c["config"], c["weight_seed"], c["codebook_decay"])
    projected = arch.codebook_dim != arch.dimension      # CostumeQuantizer.output_proj: a GEMM after the (exact) code-vector sum
    qtol = 1e-5 * float(np.sqrt((g["quantized"] ** 2).mean())) if projected else 0.0
    if rep["mismatched_indices"] == 0:
        assert rms(r["quantized"], g["quantized"]) <= qtol
        assert rms(r2["recon"], g["recon"]) < WAV_RMS_TOL
    else:
        # bit-exact, or every differing frame is PROVEN an fp32 tie of the reference's own distances (at most 1 frame in 250)
        def qin(x):      # what the residual quantiser sees: input_proj / tanh * range of the encoder output (costume_quantizer.py:84-87)
            x = torch.as_tensor(x).float().cpu()
            if projected:
                x = torch.nn.functional.linear(x, torch.from_numpy(sd["quantizer.input_proj.weight"]), torch.from_numpy(sd["quantizer.input_proj.bias"]))
            return torch.tanh(x) * arch.codec_range if arch.codec_range else x
        proofs = _assert_flips_are_near_ties(sd["quantizer.rq.model.embed"], qin(g["encoder_out"]), g["indices"].astype(np.int64), r["codes"],
                                             got_enc=qin(r["enc_out"]), max_frames=max(1, rep["frames"] // 250))
        print(f"{name}: {len(proofs)} tie frame(s) (stage, frame, margin, bound): {proofs}")
        Tf, hop = g["indices"].shape[2], m.engine.hop_length
        for b in range(c["batch"]):          # the waveform is checked regardless: whole utterances without a tie, else up to the tie
            cut = _prefix_before([p[1] for p in proofs], Tf, hop, b)
            n = c["samples"] if cut is None else min(cut, c["samples"])
            if n > 0:
                assert rms(r2["recon"][b, :, :n], g["recon"][b, :, :n]) < WAV_RMS_TOL, (b, n)
    # decode the REFERENCE's codes: isolates the decoder from any encoder-side index flip
    tok = torch.from_numpy(g["indices"].astype(np.int64)).permute(1, 2, 0).contiguous()
    w2, emb = m.engine.decode_codes(tok)
    assert rms(emb, g["quantized"]) <= qtol
    w3 = m.engine.decode_emb(torch.from_numpy(g["quantized"]))
    if "recon_from_codes" in g:
        assert rms(w2, g["recon_from_codes"]) < WAV_RMS_TOL
        assert rms(w3, g["recon_from_codes"]) < WAV_RMS_TOL
    else:                                          # un-scaled decode x the reference's scale == its scaled reconstruction
        sc = torch.from_numpy(g["scale"]).view(-1, 1, 1) if "scale" in g else 1.0
        assert rms(w2.cpu()[:, :, :c["samples"]] * sc, g["recon"]) < WAV_RMS_TOL
        assert rms(w3.cpu()[:, :, :c["samples"]] * sc, g["recon"]) < WAV_RMS_TOL


@pytest.mark.parametrize("name", SEG)
def test_segmented_mode_against_reference_golden(name):
    """model_conf.segment_dur / overlap_ratio: frames are extra batch rows for the engine, the
    triangle overlap-add runs in the reference's operation order; golden = the real reference in segmented mode."""
    from funcodec_amd.bin.codec_inference import Speech2Token   # noqa: F401  (drop-in surface is exercised elsewhere)
    c = MAN["cases"][name]
    m = engine_for(c["config"], c["weight_seed"], c["codebook_decay"])
    assert m.arch.segment_length == 8000 and m.arch.segment_stride == 7200
    # ds640seg: 8000 % 640 != 0 -> frames decode to 13 * 640 = 8320 samples; window and overlap use the untrimmed frames
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"], c.get("channels", 1))
    g = golden(name)
    r = m.inference(wav if wav.dim() == 3 else wav.unsqueeze(1), bit_width=c["bit_width"], use_scale=True)
    assert len(r["code_indices"]) == len(c["frames"]) and len(r["sub_quants"]) == len(c["frames"])
    tied = []                                          # (segment, utterance) pairs with a proven fp32 tie
    for f, idx in enumerate(r["code_indices"]):
        assert idx.shape == (c["n_q"], c["batch"], c["frames"][f])
        ref_idx = g[f"indices_{f}"].astype(np.int64)
        rep = index_report(idx, ref_idx)
        if rep["mismatched_indices"]:
            # bit-exact, or every differing frame is a PROVEN fp32 tie of the reference's own distances (as in the unsegmented tests; at most
            # one frame per segment).  The fixture stores no encoder output per segment: the CPU oracle (pinned bit-for-bit to the reference
            # on this very fixture by oracle/make_golden.py) supplies it, the engine's own comes from encoding the segment alone.
            cfg_, arch_, sd_ = state_for(c["config"], c["weight_seed"], c["codebook_decay"])
            orc = oracle_for(c["config"], c["weight_seed"], c["codebook_decay"])
            o = orc.inference(wav if wav.dim() == 3 else wav, bit_width=c["bit_width"], use_scale=True)
            # The oracle runs on THIS box's host cores: where its codes differ from the fixture (made in the build container with other
            # cores / thread counts) the reference disagrees with itself -- such a frame is not defined by "the reference" (the same rule as
            # for the committed *_variants fixtures); every other differing frame needs the tie proof.
            o_idx = o["code_indices"][f].numpy()
            ref_disagrees = (o_idx != ref_idx).any(0)                         # [B, frames]
            got = idx.cpu().numpy()
            unexplained = (got != ref_idx).any(0) & ~ref_disagrees
            assert int(ref_disagrees.sum()) <= 1
            seg = wav[..., f * m.arch.segment_stride: f * m.arch.segment_stride + m.arch.segment_length]
            own = m.engine.encode(seg, c["n_q"], want_enc_out=True)
            assert torch.equal(own["codes"], idx)
            # EVERY differing frame takes the tie proof -- also one where this box's oracle run disagrees with the fixture: such a
            # frame is a reference self-disagreement, i.e. a near-tie, and the proof (fixture code vs our code under the oracle's encoder
            # output) must then hold as well; an engine error that happened to land on it would not pass
            differing = (got != ref_idx).any(0)
            if differing.any():
                _assert_flips_are_near_ties(sd_["quantizer.rq.model.embed"], o["encoder_out"][f], ref_idx, got, got_enc=own["enc_out"], max_frames=1)
            record_report(name, segment=f, unexplained_by_oracle_rerun=int(unexplained.sum()), engine_frames_differing=np.argwhere((got != ref_idx).any(0)).tolist(),
                          oracle_on_this_box_differs_from_fixture=np.argwhere(ref_disagrees).tolist())
            bad_b = (idx.cpu().numpy() != ref_idx).any(0).any(-1)
            tied += [(f, int(b)) for b in np.nonzero(bad_b)[0]]
        assert np.allclose(r["code_embeddings"][f][1].cpu().numpy(), g[f"scale_{f}"], rtol=1e-5)
    assert r["recon_speech"].shape == (c["batch"], c.get("channels", 1), c["samples"])
    if not tied:
        assert rms(r["recon_speech"], g["recon"]) < WAV_RMS_TOL
    else:
        # a tied frame changes its whole segment of that utterance (GroupNorm statistics span the segment): every sample outside the tied
        # segments' extent must still match
        keep = torch.ones(c["batch"], c["samples"], dtype=torch.bool)
        for f, b in tied:
            keep[b, f * m.arch.segment_stride: f * m.arch.segment_stride + m.arch.segment_length] = False
        got, ref = r["recon_speech"].cpu(), torch.from_numpy(g["recon"])
        for b in range(c["batch"]):
            assert keep[b].any()
            assert rms(got[b][:, keep[b]], ref[b][:, keep[b]]) < WAV_RMS_TOL, b
    # frames of one call are independent utterances: the first frame alone gives the same codes
    one = m.engine.encode(wav[..., :8000], c["n_q"])
    assert torch.equal(one["codes"], r["code_indices"][0])


@pytest.mark.parametrize("name", [n for n, c in MAN["cases"].items() if c.get("kind") == "bypass"])
def test_bypass_quantizer_against_reference_golden(name):
    """model_conf.bypass_quantizer (codec_basic.py:148,700-705): Encodec.inference hands the encoder output on as the code embeddings with zero
    indices [B, Tf] and zero sub_quants and decodes from it (also per frame in segmented mode); inference_encoding still quantises.  Golden =
    the real reference."""
    c = MAN["cases"][name]
    m = engine_for(c["config"], c["weight_seed"])
    assert m.arch.bypass_quantizer
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"])
    g = golden(name)
    r = m.inference(wav.unsqueeze(1), bit_width=None, use_scale=True)
    assert len(r["code_indices"]) == len(c["frames"])
    for f, Tf in enumerate(c["frames"]):
        emb, scale = r["code_embeddings"][f]
        assert r["code_indices"][f].shape == (c["batch"], Tf) and r["code_indices"][f].dtype == torch.long and int(r["code_indices"][f].abs().max()) == 0
        assert r["sub_quants"][f].shape == emb.shape and float(r["sub_quants"][f].abs().max()) == 0.0
        assert rms(emb, g[f"emb_{f}"]) < 2e-5
        assert np.allclose(scale.cpu().numpy(), g[f"scale_{f}"], rtol=1e-5)
    assert r["recon_speech"].shape == g["recon"].shape and rms(r["recon_speech"], g["recon"]) < WAV_RMS_TOL
    e = m.inference_encoding(wav.unsqueeze(1), bit_width=None)
    rep = index_report(e["code_indices"][0], g["encode_indices_0"].astype(np.int64))
    assert rep["mismatched_indices"] == 0, rep


def test_stereo_model_channel_contract():
    """Stereo checkpoints (config input_size 2 / decoder_conf.channels 2; codec_basic.py:342-344,366): [B, 2, T] in, [B, 2, T] out; the volume
    scale comes from the channel MEAN, so swapping the channels keeps it; a mono tensor or a third channel is refused like the reference's
    first conv / assert would; a mono model refuses stereo input."""
    c = MAN["cases"]["tinyst_b3_t1003"]
    m = engine_for(c["config"], c["weight_seed"], c["codebook_decay"])
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"], 2)
    r = m.inference(wav)
    assert r["recon_speech"].shape == (3, 2, 1003) and r["code_indices"][0].shape == (c["n_q"], 3, c["frames"])
    sw = m.engine.encode(wav.flip(1), c["n_q"])
    assert torch.equal(sw["scale"].cpu(), r["code_embeddings"][0][1].cpu())
    assert not torch.equal(sw["codes"], r["code_indices"][0])
    with pytest.raises(ValueError):
        m.inference(wav[:, 0])                      # [B, T] -> one channel
    with pytest.raises(AssertionError):
        m.inference(torch.cat([wav, wav[:, :1]], 1))
    with pytest.raises(Exception):
        m.engine.encode(wav[:, 0], c["n_q"])
    mono = engine_for("tiny", 7)
    with pytest.raises(ValueError):
        mono.inference(wav)
    # the drop-in class takes the same tensor
    from funcodec_amd.bin.codec_inference import Speech2Token
    s2t = Speech2Token.__new__(Speech2Token)
    s2t.model, s2t.check_status, s2t.dtype = m, True, "float32"
    idx, embs, recon, subs = s2t(wav.numpy(), run_mod="inference")
    assert torch.equal(idx[0], r["code_indices"][0]) and recon.shape == (3, 2, 1003)
    tok = idx[0].permute(1, 2, 0).contiguous()
    _, _, dec, _ = s2t(tok, run_mod="decode")
    assert dec.shape == (3, 2, c["frames"] * m.engine.hop_length)


@pytest.mark.parametrize("name", ["rvq_flat", "rvq_decay08"])
def test_rvq_against_reference_golden_and_c_oracle(name):
    """32 stages x 2000 rows incl. the tie-provoking decaying codebooks: indices must be bit-exact vs the
    real reference (golden) AND vs the plain-C restatement (every row here, not a sample)."""
    import c_oracle
    from funcodec_amd.model import EncodecMI355X
    c = MAN["cases"][name]
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    sig = (c["codebook_decay"] ** np.arange(32, dtype=np.float64)).astype(np.float32)[:, None, None]
    embed = rng.standard_normal((32, 1024, 128)).astype(np.float32) * sig
    z = rng.standard_normal((8, 250, 128)).astype(np.float32) * 1.5
    cfg, arch, sd = state_for("ds640", 0)
    sd2 = dict(sd)
    sd2["quantizer.rq.model.embed"] = embed
    mm = EncodecMI355X(arch, "cuda:0")
    mm.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    codes, quant = mm.engine.rvq_encode(torch.from_numpy(z).reshape(-1, 128), 32)
    g = golden(name)
    rep = index_report(codes.reshape(32, 8, 250), g["indices"].astype(np.int64))
    assert rep["mismatched_indices"] == 0, rep
    assert rms(quant.reshape(8, 250, 128), g["quantized"]) == 0.0
    rows = slice(0, 160)
    cc, cq = c_oracle.rvq_encode(z.reshape(-1, 128)[rows], embed, 32)
    assert np.array_equal(codes.cpu().numpy()[:, rows], cc)
    assert np.array_equal(quant.cpu().numpy()[rows], cq)


def test_rvq_bit_exact_vs_c_oracle_random_shapes():
    """D=16/K=64 tiny quantiser, ragged row counts (not a multiple of the 16-row tile), planted exact ties."""
    import c_oracle
    m = engine_for("tiny", 7)
    cfg, arch, sd = state_for("tiny", 7)
    cb = sd["quantizer.rq.model.embed"]
    rng = np.random.Generator(np.random.PCG64(77))
    for N in (1, 15, 16, 17, 333):
        x = rng.standard_normal((N, 16)).astype(np.float32)
        x[0] = cb[0, 5]                       # exactly on a code vector
        codes, quant = m.engine.rvq_encode(torch.from_numpy(x), 6)
        cc, cq = c_oracle.rvq_encode(x, cb, 6)
        assert np.array_equal(codes.cpu().numpy(), cc), N
        assert np.array_equal(quant.cpu().numpy(), cq), N


def test_rvq_two_row_set_form_is_bit_exact_vs_c_oracle():
    """More than 16 rows per CU (> 4 096 rows: FreqCodec batches, 32-utterance calls) run the fused quantiser with TWO 16-row sets per
    workgroup (round 4).  Same arithmetic order per row: codes and quantised vectors bit-exact against the plain-C restatement on every
    row of a 6 000-row input (8 stages of the ds640 codebooks), and identical to the 16-row form's results on the same rows."""
    import c_oracle
    m = engine_for("ds640", 0)
    cfg, arch, sd = state_for("ds640", 0)
    cb = sd["quantizer.rq.model.embed"]
    rng = np.random.Generator(np.random.PCG64(123))
    x = (rng.standard_normal((6000, 128)) * 1.5).astype(np.float32)
    x[7] = cb[0, 11]                          # exactly on a code vector
    codes, quant = m.engine.rvq_encode(torch.from_numpy(x), 8)
    cc, cq = c_oracle.rvq_encode(x, cb, 8)
    assert np.array_equal(codes.cpu().numpy(), cc)
    assert np.array_equal(quant.cpu().numpy(), cq)
    c16, q16 = m.engine.rvq_encode(torch.from_numpy(x[:4000]), 8)       # <= 16 rows per CU: the 16-row form
    assert torch.equal(c16, codes[:, :4000]) and torch.equal(q16, quant[:4000])


@pytest.mark.parametrize("cfg_name,seed,N", [("tiny", 7, 6000), ("tinyss", 7, 5003), ("fuzz10", 10, 4700), ("fuzz2", 2, 9000)])
def test_rvq_two_row_set_form_small_codebook_dims_vs_c_oracle(cfg_name, seed, N):
    """the two-row-set form (selected above 16 rows per CU) is instantiated for D = 16 / 32 / 64 too, and only D = 128 had a
    bit-exactness test.  Codebooks of the tiny / SoundStream-shaped / pseudo-random configurations (K = 64 .. 256: waves without codes of
    their own), row counts with a ragged last workgroup, against the plain-C restatement on every row and against the 16-row form."""
    import c_oracle
    m = engine_for(cfg_name, seed)
    cfg, arch, sd = state_for(cfg_name, seed)
    cb = sd["quantizer.rq.model.embed"]
    nq = min(arch.num_quantizers, cb.shape[0])
    rng = np.random.Generator(np.random.PCG64(900 + N))
    x = (rng.standard_normal((N, arch.codebook_dim)) * 1.5).astype(np.float32)
    x[5] = cb[0, 3]
    codes, quant = m.engine.rvq_encode(torch.from_numpy(x), nq)
    cc, cq = c_oracle.rvq_encode(x, cb, nq)
    assert np.array_equal(codes.cpu().numpy(), cc)
    assert np.array_equal(quant.cpu().numpy(), cq)
    c16, q16 = m.engine.rvq_encode(torch.from_numpy(x[:3000]), nq)       # <= 16 rows per CU: the 16-row form
    assert torch.equal(c16, codes[:, :3000]) and torch.equal(q16, quant[:3000])


@pytest.mark.parametrize("cfg_name,seed,Tf,n_q", [("tinyq0", 7, 126, 6), ("tinyq0", 7, 127, 6), ("tinyq0", 7, 2, 6), ("tinyq0", 7, 3, 6),
                                                  ("ds320q0", 0, 501, 8), ("ss320q0", 0, 77, 4)])
def test_rvq_q0_ds_ratio_is_bit_exact_vs_c_oracle(cfg_name, seed, Tf, n_q):
    """quantizer_conf.q0_ds_ratio > 1 (ddp_core_vq.py:396-404) in the fused quantiser: stage 0 of frame t runs on frame
    fc_q0_source_frames(Tf)[t], every later stage on the frame's own residual.  Bit-exact against the plain-C restatement on every row
    (16-, 128- and 512-dim kernels; even / odd / minimal frame counts); the same rows through an engine WITHOUT the option differ."""
    import c_oracle
    m = engine_for(cfg_name, seed)
    cfg, arch, sd = state_for(cfg_name, seed)
    cb = sd["quantizer.rq.model.embed"]
    rng = np.random.Generator(np.random.PCG64(1000 + Tf))
    x = (rng.standard_normal((Tf, arch.codebook_dim)) * 1.5).astype(np.float32)
    codes, quant = m.engine.rvq_encode(torch.from_numpy(x), n_q)
    cc, cq = c_oracle.rvq_encode_q0(x, cb, n_q, Tf)
    assert np.array_equal(codes.cpu().numpy(), cc)
    assert np.array_equal(quant.cpu().numpy(), cq)
    if Tf > 3:
        plain, _ = c_oracle.rvq_encode(x, cb, n_q)
        assert not np.array_equal(plain, cc)


def test_use_ddp_false_checkpoint_layout_against_reference_golden():
    """`use_ddp: false` checkpoints store one codebook per layer (`quantizer.rq.model.layers.{i}._codebook.embed`,
    core_vq.py:147-150,324-396).  Golden = core_vq.ResidualVectorQuantization itself, run in the build container."""
    from funcodec_amd.config import arch_from_config, recipe_config
    from funcodec_amd.model import EncodecMI355X
    from funcodec_amd.synth import make_state_dict
    c = MAN["cases"]["rvq_noddp"]
    nq = c["n_q"]
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    embed = rng.standard_normal((nq, 1024, 128)).astype(np.float32)
    z = rng.standard_normal((4, 125, 128)).astype(np.float32) * 1.5
    cfg = recipe_config("ds640")
    cfg["quantizer_conf"]["num_quantizers"] = nq
    cfg["quantizer_conf"]["use_ddp"] = False
    arch = arch_from_config(cfg)
    sd = {k: v for k, v in make_state_dict(arch, 0).items() if not k.startswith("quantizer.")}
    for i in range(nq):
        sd[f"quantizer.rq.model.layers.{i}._codebook.embed"] = embed[i]
        sd[f"quantizer.rq.model.layers.{i}._codebook.inited"] = np.ones((1,), np.float32)
    mm = EncodecMI355X(arch, "cuda:0")
    mm.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    codes, quant = mm.engine.rvq_encode(torch.from_numpy(z).reshape(-1, 128), nq)
    g = golden("rvq_noddp")
    rep = index_report(codes.reshape(nq, 4, 125), g["indices"].astype(np.int64))
    assert rep["mismatched_indices"] == 0, rep
    assert rms(quant.reshape(4, 125, 128), g["quantized"]) == 0.0


def test_overlap_add_entry_point_against_the_reference_formula():
    """fc_overlap_add = _linear_overlap_add (codec_basic.py:77-116): ragged last frame, more than two frames per position
    (stride < length / 2), a single frame, output trimmed to a shorter length."""
    import torch_oracle as TO
    m = engine_for("tiny", 7)
    gen = torch.Generator().manual_seed(3)
    for lens, stride, out_len in (([832, 832, 500], 720, 1900), ([400, 400, 400, 273, 123], 150, None), ([77], 5, None),
                                  ([8320, 8320, 5760], 7200, 20000)):
        frames = [torch.randn(3, 1, n, generator=gen) for n in lens]
        ref = TO.Oracle.linear_overlap_add(frames, stride)
        if out_len is not None:
            ref = ref[..., :out_len]
        got = m.engine.overlap_add([f.cuda() for f in frames], stride, out_len).cpu()
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() < 2e-6, (lens, stride)


def test_deferred_device_errors_are_reported():
    """Kernels cannot return a status: an out-of-range code index (the reference's F.embedding raises, ddp_core_vq.py:191)
    and a persistent-LSTM grid-barrier timeout are written to host-visible status words and raised by check_status() /
    the next engine call; afterwards the engine keeps working (the LSTM on its per-step launch path, same bits)."""
    import subprocess
    import sys
    from funcodec_amd.engine import EngineError
    m = engine_for("ds320", 0)
    eng = m.engine
    wav = audio(2, 6400, 3, "tones").cuda()
    good = eng.encode_decode(wav, 32)
    eng.check_status()                                               # nothing pending
    tok = good["codes"].permute(1, 2, 0).contiguous().clone()
    tok[0, 0, 0] = 99999
    w, _ = eng.decode_codes(tok)
    assert torch.isfinite(w).all()                                    # clamped, never an out-of-bounds read
    with pytest.raises(EngineError, match="outside"):
        eng.check_status()
    eng.check_status()                                               # reported once
    tok[0, 0, 0] = -3
    eng.decode_codes(tok)
    torch.cuda.synchronize()
    with pytest.raises(EngineError, match="outside"):                 # ... or by the next compute call
        eng.encode_decode(wav, 32)
    again = eng.encode_decode(wav, 32)
    assert torch.equal(again["codes"], good["codes"]) and torch.equal(again["recon"], good["recon"])
    # forced barrier "timeout" (FC_ABLATE_LSTM=64 is the kernel's test hook) in a subprocess: env is read once per process
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')\n"
        "from helpers import engine_for, audio\n"
        "from funcodec_amd.engine import EngineError\n"
        "m = engine_for('ds320', 0); wav = audio(2, 6400, 3, 'tones').cuda()\n"
        "r = m.engine.encode_decode(wav, 32)\n"
        "try:\n"
        "    m.engine.check_status(); raise SystemExit('timeout was not reported')\n"
        "except EngineError as e:\n"
        "    assert 'grid barrier' in str(e), str(e)\n"
        "assert not torch.isfinite(r['recon']).all()\n"
        "r2 = m.engine.encode_decode(wav, 32); m.engine.check_status()\n"     # per-step fallback from now on
        "assert torch.isfinite(r2['recon']).all()\n"
        "torch.save(r2['codes'].cpu(), sys.argv[1])\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "_status_codes.pt")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    subprocess.run([sys.executable, "-c", code, path], check=True, cwd=root, env=dict(os.environ, FC_ABLATE_LSTM="64"), timeout=300)
    assert torch.equal(torch.load(path), good["codes"].cpu())


# ---- (b) per-op parity against torch.nn.functional on CPU -----------------------------------------
def _layer_cases(cfg_name, seed):
    cfg, arch, sd = state_for(cfg_name, seed)
    out = []
    for k in sd:
        for suffix in (".conv.bias", ".convtr.bias"):
            if k.endswith(suffix) and k.startswith(("encoder.", "decoder.")):
                out.append(k[: -len(suffix)])
    return sorted(set(out))


@pytest.mark.parametrize("cfg_name,seed,B,T0", [("tiny", 7, 3, 203), ("ds640", 0, 2, 3200)])
def test_every_conv_layer_against_torch_cpu(cfg_name, seed, B, T0):
    import torch_oracle as TO
    m = engine_for(cfg_name, seed)
    orc = oracle_for(cfg_name, seed)
    gen = torch.Generator().manual_seed(5)
    worst = 0.0
    for p in _layer_cases(cfg_name, seed):
        tr = p.endswith("convtr")
        w = orc.sd[p + (".convtr.weight" if tr else ".conv.weight")]
        cin, k = (w.shape[0] if tr else w.shape[1]), w.shape[2]
        T = max(3, T0 // max(1, cin // 8)) if cfg_name == "tiny" else max(7, 4 * T0 // cin)
        x = torch.randn(B, cin, T, generator=gen)
        for elu in (False, True):
            xin = F.elu(x) if elu else x
            if tr:
                ref = TO.sconvtr1d(xin, *orc._p(p), k // 2, orc.eps)
            else:
                ref = TO.sconv1d(xin, *orc._p(p), (k // 2 if (k % 2 == 0 and k > 1) else 1), orc.eps)
            got = m.engine.layer_forward(p, x, apply_elu=elu).cpu()
            assert got.shape == ref.shape, (p, got.shape, ref.shape)
            err = (got - ref).abs().max().item()
            worst = max(worst, err)
            assert err < LAYER_ABS_TOL, (p, elu, err)
    print(f"worst single-layer abs err {worst:.2e}")


@pytest.mark.parametrize("T", [1, 2, 3, 5, 8, 64, 129, 257])
def test_conv_padding_edge_lengths(T):
    """reflect padding on inputs shorter than the pad (pad1d zero-extension, conv.py:89-97), odd lengths that
    need 'extra' right padding, lengths straddling the 128/256-column tiles."""
    import torch_oracle as TO
    m = engine_for("tiny", 7)
    orc = oracle_for("tiny", 7)
    gen = torch.Generator().manual_seed(T)
    for p, stride in (("encoder.model.0.conv", 1), ("encoder.model.3.conv", 2), ("encoder.model.6.conv", 4),
                      ("encoder.model.1.block.1.conv", 1), ("decoder.model.9.conv", 1)):
        w = orc.sd[p + ".conv.weight"]
        x = torch.randn(2, w.shape[1], T, generator=gen)
        ref = TO.sconv1d(x, *orc._p(p), stride, orc.eps)
        got = m.engine.layer_forward(p, x).cpu()
        assert got.shape == ref.shape
        # GroupNorm over very few elements amplifies rounding; compare with a relative floor
        assert (got - ref).abs().max().item() < 2e-4, (p, T)
    for p in ("decoder.model.3.convtr", "decoder.model.6.convtr"):
        w = orc.sd[p + ".convtr.weight"]
        x = torch.randn(2, w.shape[0], T, generator=gen)
        ref = TO.sconvtr1d(x, *orc._p(p), w.shape[2] // 2, orc.eps)
        got = m.engine.layer_forward(p, x).cpu()
        assert got.shape == ref.shape and (got - ref).abs().max().item() < 2e-4, (p, T)


def test_conv_layers_random_shape_sweep():
    """Seeded sweep over (layer, batch, length, ELU) of the real recipes (GroupNorm, weight-norm causal, SoundStream shape): lengths around the tile widths (127..129,
    255..257, 1023..1025), tiny lengths, primes; every staging scheme (element / row / single-output-channel kernel), edge-
    only and interior tiles, strided and transposed layers.  Reference = the torch restatement of SConv1d / SConvTranspose1d."""
    import torch_oracle as TO
    rng = np.random.Generator(np.random.PCG64(2024))
    lengths = [1, 2, 5, 17, 63, 127, 128, 129, 255, 256, 257, 511, 640, 1023, 1024, 1025, 1531, 2053]
    from funcodec_amd.plan import decoder_plan, encoder_plan
    for cfg_name in ("ds320", "ds640", "ds320wn", "ss320"):
        m = engine_for(cfg_name, 0)
        orc = oracle_for(cfg_name, 0)
        layers = _layer_cases(cfg_name, 0)
        dil = {op.key: op.dilation for op in encoder_plan(m.arch) + decoder_plan(m.arch)}
        for _ in range(36):
            p = layers[int(rng.integers(len(layers)))]
            tr = p.endswith("convtr")
            w = orc._p(p)[0]                         # (folded) weight: [Cout,Cin,k] or [Cin,Cout,k]
            cin, k = (w.shape[0] if tr else w.shape[1]), w.shape[2]
            T = lengths[int(rng.integers(len(lengths)))]
            if cin >= 512:
                T = min(T, 257)                      # keep the CPU reference cheap on the wide layers
            B = int(rng.integers(1, 4))
            elu = bool(rng.integers(2))
            x = torch.from_numpy(rng.standard_normal((B, cin, T)).astype(np.float32))
            xin = F.elu(x) if elu else x
            if tr:
                ref = TO.sconvtr1d(xin, *orc._p(p), k // 2, orc.eps, orc.causal)
            else:
                ref = TO.sconv1d(xin, *orc._p(p), (k // 2 if (k % 2 == 0 and k > 1) else 1), orc.eps, orc.causal, dil[p])
            got = m.engine.layer_forward(p, x, apply_elu=elu).cpu()
            assert got.shape == ref.shape, (cfg_name, p, B, T, got.shape, ref.shape)
            # GroupNorm over very few elements (short T) amplifies rounding: tolerance as in the padding edge test
            tol = 2e-4 if ref.shape[-1] * ref.shape[1] < 4096 else LAYER_ABS_TOL
            err = (got - ref).abs().max().item()
            assert err < tol, (cfg_name, p, B, T, elu, err)


@pytest.mark.parametrize("cfg_name", ["ds640", "ss320nc", "ss320"])
def test_fused_resblock_head_against_torch_cpu(cfg_name):
    """The thin residual blocks (C = 32 / 64) run shortcut + block.1 as ONE launch (reshead_kernel): whole blocks against the
    torch restatement of SEANetResnetBlock.forward (seanet_encoder.py:44-61) at lengths around the 128-column tile (edge-only,
    interior, straddling, shorter than the reflect pad), GroupNorm / weight-norm causal flavours and dilations 1, 2, 4."""
    from funcodec_amd.plan import decoder_plan, encoder_plan
    m = engine_for(cfg_name, 0)
    orc = oracle_for(cfg_name, 0)
    ops = [op for op in encoder_plan(m.arch) + decoder_plan(m.arch) if op.key.endswith(".block.1.conv") and op.cin in (32, 64)]
    assert ops
    gen = torch.Generator().manual_seed(11)
    worst = 0.0
    for op in ops:
        prefix = op.key[: -len(".block.1.conv")]
        for B, T in ((2, 1), (1, 3), (3, 127), (2, 128), (2, 129), (1, 257), (2, 1000), (1, 4099)):
            x = torch.randn(B, op.cin, T, generator=gen)
            ref = orc._resblock(x, prefix, op.dilation)
            got = m.engine.resblock_forward(prefix, x).cpu()
            assert got.shape == ref.shape
            err = (got - ref).abs().max().item()
            worst = max(worst, err)
            tol = 4e-4 if T * op.cin < 4096 else 2 * LAYER_ABS_TOL      # GroupNorm over very few elements amplifies rounding
            assert err < tol, (cfg_name, prefix, B, T, err)
    print(f"{cfg_name}: worst fused res-block abs err {worst:.2e} over {len(ops)} blocks")


@pytest.mark.parametrize("T", [769, 1024, 1291])
def test_row_staging_interior_and_straddling_tiles(T):
    """The stride-1 layers of the real recipe use row staging (16-byte loads, one channel row per 32 / 64 lanes, k-1 tail
    columns per thread): lengths with interior tiles, a last tile that is exactly full, and one that straddles; k = 1, 3,
    7 and the 2-tap transposed-conv GEMM; 32-, 64- and 128-row tiles."""
    import torch_oracle as TO
    m = engine_for("ds640", 0)
    orc = oracle_for("ds640", 0)
    gen = torch.Generator().manual_seed(T)
    for p in ("encoder.model.1.shortcut.conv", "encoder.model.1.block.1.conv", "encoder.model.4.block.3.conv",
              "decoder.model.13.block.1.conv", "decoder.model.18.conv", "decoder.model.10.shortcut.conv"):
        w = orc.sd[p + ".conv.weight"]
        x = torch.randn(2, w.shape[1], T, generator=gen)
        for elu in (False, True):
            ref = TO.sconv1d(F.elu(x) if elu else x, *orc._p(p), 1, orc.eps)
            got = m.engine.layer_forward(p, x, apply_elu=elu).cpu()
            assert got.shape == ref.shape and (got - ref).abs().max().item() < LAYER_ABS_TOL, (p, T, elu)
    for p in ("decoder.model.15.convtr", "decoder.model.12.convtr"):
        w = orc.sd[p + ".convtr.weight"]
        x = torch.randn(2, w.shape[0], T // 2, generator=gen)
        ref = TO.sconvtr1d(x, *orc._p(p), w.shape[2] // 2, orc.eps)
        got = m.engine.layer_forward(p, x).cpu()
        assert got.shape == ref.shape and (got - ref).abs().max().item() < LAYER_ABS_TOL, (p, T)


def test_staging_scheme_and_workgroup_count_do_not_change_results():
    """FC_TARGET_WGS only changes how many N tiles a workgroup walks: results must be bit-identical.  FC_ROW=0 (element
    staging everywhere, other K chunking) changes the summation order across chunks: same codes, waveform within tolerance."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, os, torch, numpy as np\n"
        "sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')\n"
        "from helpers import engine_for, audio\n"
        "m = engine_for('ds320', 0)\n"
        "r = m.engine.encode_decode(audio(3, 24000, 11, 'tones').cuda(), 32)\n"
        "np.savez(sys.argv[1], codes=r['codes'].cpu().numpy(), recon=r['recon'].cpu().numpy())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, env in (("base", {}), ("fewwg", {"FC_TARGET_WGS": "96"}), ("norow", {"FC_ROW": "0"})):
        path = os.path.join(root, "gpurun_out", f"_variant_{tag}.npz")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.run([sys.executable, "-c", code, path], check=True, cwd=root, env=dict(os.environ, **env), timeout=300)
        outs[tag] = np.load(path)
    assert np.array_equal(outs["base"]["codes"], outs["fewwg"]["codes"])
    assert np.array_equal(outs["base"]["recon"], outs["fewwg"]["recon"])
    assert np.array_equal(outs["base"]["codes"], outs["norow"]["codes"])
    assert rms(outs["base"]["recon"], outs["norow"]["recon"]) < WAV_RMS_TOL


@pytest.mark.parametrize("cfg_name,seed,B,T", [("tiny", 7, 3, 9), ("tiny", 7, 17, 4), ("ds640", 0, 2, 9), ("ds320", 0, 33, 3),
                                                ("ds640", 0, 32, 7), ("ds320", 0, 20, 11)])     # two batch tiles in the persistent kernel
def test_lstm_against_torch_cpu(cfg_name, seed, B, T):
    m = engine_for(cfg_name, seed)
    orc = oracle_for(cfg_name, seed)
    for p in [k[: -len(".weight_ih_l0")] for k in orc.sd if k.endswith(".weight_ih_l0")]:
        H = orc.sd[p + ".weight_ih_l0"].shape[1]
        x = torch.randn(B, H, T, generator=torch.Generator().manual_seed(6))
        ref = orc._slstm(x, p)
        got = m.engine.lstm_forward(p, x).cpu()
        assert (got - ref).abs().max().item() < 1e-5, p


def test_engine_calls_can_be_replayed_as_a_hip_graph():
    """fc_* calls only enqueue kernels of this library on the given stream, so a caller may stream-capture them (torch.cuda.CUDAGraph).
    Round 2: the barrier words / initial LSTM state used to be cleared with hipMemsetAsync, whose graph memset node replayed unordered with
    its neighbours (persistent-LSTM barrier timeouts, wrong states); they are cleared by a kernel of the library now."""
    m = engine_for("ds320", 0)
    wav = audio(3, 24000, 5, "tones").cuda()
    ref = m.engine.encode_decode(wav, 32)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.engine.encode_decode(wav, 32)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m.engine.encode_decode(wav, 32)
    for i in range(40):
        g.replay()
        if i % 13 == 0:
            torch.cuda.synchronize()
            assert torch.equal(out["codes"], ref["codes"]) and torch.equal(out["recon"], ref["recon"]), i
    torch.cuda.synchronize()
    assert torch.equal(out["codes"], ref["codes"]) and torch.equal(out["recon"], ref["recon"])
    m.engine.check_status()


def test_two_engines_on_two_threads_and_streams_concurrently():
    """Two engines driven from two Python threads on two HIP streams at once (conv kernels of one next to the persistent LSTM of the other,
    thread-local error strings, per-engine status words): every call still returns the single-stream result bit for bit."""
    import threading
    from funcodec_amd.model import EncodecMI355X
    cfg, arch, sd = state_for("ds640", 0)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    wavs = [audio(8, 48000, 11 + i, "tones").cuda() for i in range(2)]
    m0 = engine_for("ds640", 0)
    refs = [m0.engine.encode_decode(w, 32) for w in wavs]
    torch.cuda.synchronize()
    errs = []

    def worker(i):
        try:
            m = EncodecMI355X(arch, "cuda:0")
            m.load_state_dict(tsd)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for it in range(12):
                    r = m.engine.encode_decode(wavs[i], 32)
                    if not (torch.equal(r["codes"], refs[i]["codes"]) and torch.equal(r["recon"], refs[i]["recon"])):
                        errs.append((i, it, "mismatch"))
                s.synchronize()
                m.engine.check_status()
        except Exception as ex:                      # noqa: BLE001
            errs.append((i, repr(ex)[:300]))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:3]


# ---- long recurrences and degenerate signals ---------------------------------------------------------------------------------------
def test_long_utterance_against_oracle():
    """60 s in one utterance (ds320: a 3 000-step LSTM recurrence on hardware exp2 / rcp gates, 3 750 tiles on the thin layers, 32-bit
    offsets): the indices still match the CPU path (any flip proven a tie) and the waveform stays within 1e-4 RMS."""
    m, orc = engine_for("ds320", 0), oracle_for("ds320", 0)
    wav = audio(1, 16000 * 60, 7, "tones")
    o = orc.inference(wav, bit_width=None, use_scale=True)
    r = m.engine.encode_decode(wav.cuda(), 32)
    m.engine.check_status()
    assert r["codes"].shape[2] == 3000
    rep = index_report(r["codes"], o["code_indices"][0])
    if rep["frames_bad"]:
        proofs = _assert_flips_are_near_ties(orc.embed, o["encoder_out"], o["code_indices"][0], r["codes"], max_frames=3)
        cut = _prefix_before([p[1] for p in proofs], 3000, m.engine.hop_length, 0)
        assert cut is None or rms(r["recon"][0, :, :cut], o["recon_speech"][0, :, :cut]) < WAV_RMS_TOL
    else:
        assert rms(r["recon"], o["recon_speech"]) < WAV_RMS_TOL


@pytest.mark.parametrize("cfg_name,seed", [("ds320", 0), ("tinywn", 9), ("tinyfreq", 3)])
def test_degenerate_signals_against_oracle(cfg_name, seed):
    """Digital silence (volume 0 -> scale 1e-8, GroupNorm over constant tensors), a full-scale clipped square wave, a DC offset and a
    single click: the reference path's behaviour on them is reproduced (the reference has no special cases for them either)."""
    freq = cfg_name.startswith("tinyfreq")
    if freq:
        from helpers import freq_engine_for, freq_oracle_for
        m, orc = freq_engine_for(cfg_name, seed), freq_oracle_for(cfg_name, seed)
    else:
        m, orc = engine_for(cfg_name, seed), oracle_for(cfg_name, seed)
    T = 4000
    t = torch.arange(T, dtype=torch.float32)
    click = torch.zeros(T)
    click[1234] = 0.9
    wav = torch.stack([torch.zeros(T), torch.sign(torch.sin(2 * np.pi * 220.0 * t / 16000.0)), torch.full((T,), 0.5), click])
    if freq:
        # STFT-domain codec: log|X| and X / |X| of bins that are ANALYTICALLY zero (a pure DC offset, the even harmonics of a square wave)
        # are functions of the transform's rounding noise, which no two FFT implementations share -- keep the rows whose spectrum is
        # either exactly zero (silence: the clamp decides) or nowhere zero (a click)
        wav = torch.stack([wav[0], wav[3], wav[0], wav[3]])
    o = orc.inference(wav, bit_width=None, use_scale=True)
    ret = m.inference(wav.cuda().unsqueeze(1), bit_width=None, use_scale=True)
    m.engine.check_status()
    assert bool(torch.isfinite(ret["recon_speech"]).all()) == bool(torch.isfinite(o["recon_speech"]).all())
    for b in range(4):                        # per utterance: a degenerate row may sit on exact ties of its own
        rep = index_report(ret["code_indices"][0][:, b:b + 1], o["code_indices"][0][:, b:b + 1])
        ref_rms = max(float(o["recon_speech"][b].double().pow(2).mean().sqrt()), 1e-3)
        if rep["frames_bad"] == 0:
            assert rms(ret["recon_speech"][b], o["recon_speech"][b]) < 1e-3 * ref_rms, b
        else:
            _assert_flips_are_near_ties(orc.embed, o["encoder_out"][b:b + 1], o["code_indices"][0][:, b:b + 1], ret["code_indices"][0][:, b:b + 1],
                                        max_frames=max(2, rep["frames"] // 4))


# ---- pseudo-random architectures (config.py::fuzz_recipe_config; five more of them have goldens from the real reference above) -------
@pytest.mark.parametrize("seed", [1, 4, 5, 6, 7, 8, 9, 12, 13, 14,
                                  3001, 3002, 3007, 3012, 3019])     # >= 3000 also draw stereo models / q0_ds_ratio (60-seed sweep clean on MI355X)
def test_random_architectures_against_oracle(seed):
    m, orc = engine_for(f"fuzz{seed}", seed), oracle_for(f"fuzz{seed}", seed)
    B, T = 1 + seed % 3, 1500 + 377 * (seed % 100)
    ch = m.arch.input_channels
    wav = audio(B, T, 3000 + seed, "tones" if seed % 2 else "noise", ch)
    o = orc.inference(wav, bit_width=None, use_scale=True)
    ret = m.inference(wav.cuda() if ch > 1 else wav.cuda().unsqueeze(1), bit_width=None, use_scale=True)
    m.engine.check_status()
    rep = index_report(ret["code_indices"][0], o["code_indices"][0])
    if rep["frames_bad"]:
        _assert_flips_are_near_ties(orc.embed, o["encoder_out"], o["code_indices"][0], ret["code_indices"][0], max_frames=max(1, rep["frames"] // 100))
    else:
        assert rms(ret["recon_speech"], o["recon_speech"]) < WAV_RMS_TOL
        assert rms(ret["code_embeddings"][0][0], o["code_embeddings"][0][0]) == 0.0
    tok = o["code_indices"][0].permute(1, 2, 0).contiguous()
    assert rms(m.engine.decode_codes(tok)[0], orc.decode_codes(tok)[0]) < WAV_RMS_TOL


# ---- oracle on fresh seeded inputs (sizes the CPU finishes in seconds) ------------------------------
@pytest.mark.parametrize("cfg_name,seed,decay,B,T,kind,bw", [
    ("tiny", 9, 1.0, 4, 4001, "tones", None),
    ("tiny", 9, 0.8, 2, 777, "noise", 4000),
    ("ds320", 3, 1.0, 2, 24000, "tones", None),
    ("ds640", 4, 1.0, 3, 32000, "noise", None),
    ("ds640", 4, 0.8, 1, 48001, "tones", 2000),
])
def test_e2e_against_oracle_fresh_inputs(cfg_name, seed, decay, B, T, kind, bw):
    m = engine_for(cfg_name, seed, decay)
    orc = oracle_for(cfg_name, seed, decay)
    wav = audio(B, T, 1000 + T, kind)
    o = orc.inference(wav, bit_width=bw, use_scale=True)
    ret = m.inference(wav.cuda().unsqueeze(1), bit_width=bw, use_scale=True)
    # Index parity vs the CPU path: the encoder outputs agree to ~1e-6, which can flip a near-tie.  A flipped frame
    # must be a PROVEN near-tie of the oracle's own distances, and the waveform is checked regardless (up to the tie).
    enc = m.engine.encode(wav.cuda(), o["code_indices"][0].shape[0], want_enc_out=True)
    got = dict(codes=ret["code_indices"][0], recon=ret["recon_speech"], enc_out=enc["enc_out"])
    proofs = _check_against(orc.embed, got, o["code_indices"][0], o["encoder_out"], o["recon_speech"], m.engine.hop_length)
    assert ret["sub_quants"][0].shape == o["sub_quants"][0].shape
    if not proofs:
        assert rms(ret["code_embeddings"][0][0], o["code_embeddings"][0][0]) == 0.0
        assert rms(ret["sub_quants"][0], o["sub_quants"][0]) == 0.0


def _assert_flips_are_near_ties(embed, ref_enc, ref_idx, got_idx, got_enc=None, max_frames=None):
    """Index parity against ONE CPU run is only defined up to fp32 ties: the reference's own encoder output moves by ~1e-6
    with its thread count / BLAS build, and the distance -(|x|^2 - 2x.e + |e|^2) is evaluated with
    cancellation at |dist| ~ 200-400 (ulp 1.5e-5 ... 3e-5).  A differing frame is accepted ONLY if, at its FIRST divergent
    stage, the reference's own margin between the two codes is below what the measured perturbation can move it:
        gap = dist[ref code] - dist[our code]  <=  2 |e_ref - e_ours| |x_ours - x_ref|  +  1e-6 max|dist|
    (first-order effect of the encoder-output difference of THAT frame + rounding of the distance evaluation itself;
    without our encoder output: 5e-6 max|dist|).  Later stages of such a frame see another residual and are not comparable.
    Returns [(stage, frame, gap, bound)]."""
    embed = torch.as_tensor(embed).float().cpu()
    ref_enc = torch.as_tensor(ref_enc).float().cpu()
    ref_idx = torch.as_tensor(ref_idx).long().cpu()
    got_idx = torch.as_tensor(got_idx).long().cpu()
    nq = ref_idx.shape[0]
    resid = ref_enc.reshape(-1, ref_enc.shape[-1]).clone()
    dx = None if got_enc is None else (torch.as_tensor(got_enc).float().cpu().reshape(resid.shape) - resid).norm(dim=1)
    g2, r2 = got_idx.reshape(nq, -1), ref_idx.reshape(nq, -1)
    bad_frames = (g2 != r2).any(0).nonzero().flatten().tolist()
    first = {n: int((g2[:, n] != r2[:, n]).float().argmax()) for n in bad_frames}
    proofs = []
    for i in range(nq):
        e = embed[i]
        todo = [n for n in bad_frames if first[n] == i]
        if todo:
            x = resid[todo]
            dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ e.t() + e.pow(2).sum(1)[None])
            for j, n in enumerate(todo):
                a, b = int(r2[i, n]), int(g2[i, n])
                gap = float(dist[j, a] - dist[j, b])
                scale = float(dist[j].abs().max())
                bound = 5e-6 * scale if dx is None else 2.0 * float((e[a] - e[b]).norm()) * float(dx[n]) + 1e-6 * scale
                proofs.append((i, n, gap, bound))
                assert -bound <= gap <= bound, f"stage {i} frame {n}: index {b} instead of {a}, margin {gap:.3e} > {bound:.3e}: NOT a tie"
        resid = resid - e[r2[i]]
    if max_frames is not None:
        assert len(bad_frames) <= max_frames, (len(bad_frames), proofs)
    # every accepted tie goes on record (conftest.py: summary line + gpurun_out/tie_waivers.json)
    record_waivers(os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::")[-1], proofs)
    return proofs


def _prefix_before(flip_frames, Tf, hop, b):
    """Samples of utterance b that precede every flipped frame's neighbourhood (the decoder's LSTM runs forward in time and
    its convolutions reach a few frames back): the reconstruction must still match there."""
    ts = [n - b * Tf for n in flip_frames if b * Tf <= n < (b + 1) * Tf]
    return None if not ts else max(0, (min(ts) - 8) * hop)


# ---- FreqCodec: STFT-domain codec over the 2-D SEANet -------------------------
@pytest.mark.parametrize("name", FREQ)
def test_freq_codec_against_reference_golden(name):
    """FreqCodec.inference / inference_decoding / inference_decoding_emb (codec_freq.py:668-834) against the real reference's
    outputs: bit-exact indices and quantised embeddings, waveform within 1e-4 RMS; the reconstruction is as long as the
    reference's (shorter than the input when the utterance has an even number of STFT frames)."""
    from helpers import freq_engine_for, freq_state_for
    c = MAN["cases"][name]
    m = freq_engine_for(c["config"], c["weight_seed"])
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"], c.get("channels", 1))
    g = golden(name)
    r = m.engine.encode(wav, c["n_q"], want_enc_out=True)
    # `stft_self_noise` (MANIFEST.json, measured by oracle/make_golden.py): how far the REFERENCE's own encoder output moves when its
    # fp32 FFT is replaced by the exact transform.  Bins near the FFT's rounding floor carry phases made of rounding noise (10 % of the
    # bins of the band-limited jamendo recording sit below 1e-3: 5.7e-4; speech and synthetic cases: 1e-6 .. 3e-5), so a fixture pins
    # any second implementation only down to that number.
    # with CostumeQuantizer's projection the quantised embedding leaves a GEMM (output_proj): float-close instead of bit-exact
    qtol = 1e-5 * float(np.sqrt((g["quantized"] ** 2).mean())) if m.arch.codebook_dim != m.arch.dimension else 0.0
    noise = float(c.get("stft_self_noise", 0.0))
    assert rms(r["enc_out"], g["encoder_out"]) < max(1e-4, 2.0 * noise)
    ill = noise > 1e-4
    ref = g["indices"].astype(np.int64)                                  # [nq, B, Tf]
    ours = r["codes"].cpu().numpy()
    # COMMITTED FACTS about the fixture (oracle/make_golden.py, `<name>_variants.npz`): the REAL reference with one thread, three threads and
    # with its STFT evaluated exactly (fp64).  Frames on which those runs disagree with the fixture are not defined by "the reference".
    variants = {}
    if name + "_variants" in MAN["cases"]:
        v = golden(name + "_variants")
        variants = {k[len("indices_"):]: v[k].astype(np.int64) for k in v if k.startswith("indices_")}
    ref_disagrees = np.zeros(ref.shape[1:], dtype=bool)                   # [B, Tf]
    for iv in variants.values():
        ref_disagrees |= (iv != ref).any(0)
    ours_differs = (ours != ref).any(0)
    if "scale" in g:
        assert float(((r["scale"].cpu() - torch.from_numpy(g["scale"])).abs() / torch.from_numpy(g["scale"])).max()) < 1e-5
    else:
        assert r.get("scale") is None and not m.arch.audio_normalize
    embed = freq_state_for(c["config"], c["weight_seed"])[2]["quantizer.rq.model.embed"]
    if ill:
        # band-limited music: 10 % of the STFT bins sit at the fp32 FFT's rounding floor, their phase features are rounding noise and the
        # encoder output of ANY second transform moves by `stft_self_noise`.  How far the codes move under such a perturbation is a fact of
        # the fixture: the reference itself with the exact STFT (variant "stft64").  The engine is held to that: per stage, its codes are
        # compared with the SET of reference runs, the numbers go on record, and it may not scatter more than a few times what the
        # reference's own variant does.
        runs = dict(fixture=ref, **variants)
        in_set = np.zeros_like(ours, dtype=bool)
        for iv in runs.values():
            in_set |= ours == iv
        per_stage_set = in_set.reshape(ours.shape[0], -1).mean(1)
        frames_equal_some_run = max(int((ours == iv).all(0).sum()) for iv in runs.values())
        frames_total = int(ours_differs.size)
        ref_self = int(ref_disagrees.sum())
        facts = dict(frames=frames_total, reference_self_disagreement_frames=ref_self,
                     engine_frames_differing_from_fixture=int(ours_differs.sum()),
                     engine_frames_identical_to_best_single_run=frames_equal_some_run,
                     first_stage_agreement_with_fixture=float((ours[0] == ref[0]).mean()),
                     first_stage_agreement_with_set=float(per_stage_set[0]),
                     per_stage_agreement_with_set=[round(float(x), 4) for x in per_stage_set],
                     all_stage_agreement_with_fixture=float((ours == ref).mean()),
                     reference_variants_all_stage_agreement={k: float((iv == ref).mean()) for k, iv in variants.items()},
                     encoder_out_rms_vs_fixture=rms(r["enc_out"], g["encoder_out"]), stft_self_noise=noise)
        record_report(name, **facts)
        print(f"{name}: ill-conditioned STFT fixture: {facts}")
        assert variants, "the ill-conditioned fixture needs its committed reference variants"
        assert facts["first_stage_agreement_with_set"] >= 0.99, facts
        # the reference's own exact-STFT variant moves `ref_self` frames; the engine may not scatter more than that + 2 frames from the best single
        # reference run (measured: 6 against ref_self = 7; rounds 4-5 allowed max(8, 4 * ref_self) = 28, which was a tolerance, not a pin)
        assert frames_total - frames_equal_some_run <= ref_self + 2, facts
        # what follows (decode path) runs from the REFERENCE's codes
        r = dict(r, codes=torch.from_numpy(ref).to(r["codes"].device), quantized=torch.from_numpy(g["quantized"]))
    else:
        unexplained = ours_differs & ~ref_disagrees
        if unexplained.any():
            # not a frame the reference disagrees with itself on: it must be a PROVEN fp32 tie of the reference's own distances
            keep = torch.from_numpy(np.where(unexplained[None], ours, ref))        # ours on the unexplained frames, the fixture elsewhere
            _assert_flips_are_near_ties(embed, g["encoder_out"], ref, keep, got_enc=r["enc_out"], max_frames=1)
        if ours_differs.any():
            record_report(name, engine_frames_differing_from_fixture=np.argwhere(ours_differs).tolist(),
                          reference_self_disagreement_frames=np.argwhere(ref_disagrees).tolist(),
                          explained_by_reference_variants=int((ours_differs & ref_disagrees).sum()),
                          encoder_out_rms_vs_fixture=rms(r["enc_out"], g["encoder_out"]),
                          reference_variants_encoder_out_rms=MAN["cases"].get(name + "_variants", {}).get("summary", {}) and
                          {k: v["encoder_out_rms_diff"] for k, v in MAN["cases"][name + "_variants"]["summary"].items()})
        else:
            assert rms(r["quantized"], g["quantized"]) <= qtol
    r2 = m.engine.encode_decode(wav, c["n_q"], use_scale=True)
    m.engine.check_status()
    assert ill or torch.equal(r2["codes"], r["codes"])
    assert tuple(r2["recon"].shape) == g["recon"].shape                 # (B, 1, min(T, decoded samples))
    # the waveform is checked whether or not a frame flipped, against the SIGNAL's level (the recordings are quiet: 1e-4 absolute would be
    # 3 % of libritts_8230's RMS): 1e-3 of the reference reconstruction's RMS without a flip.  With ONE flipped frame elsewhere the
    # GroupNorm(1, C) statistics of every decoder layer span the whole utterance, so the samples before the flip move by O(1 / frames) of
    # the signal: the bar there is 2 / frames of the RMS (libritts_8230: 92 frames -> 2.2e-2; measured 1.01e-2), never below 1e-2
    flips = np.nonzero(ours_differs.reshape(-1))[0].tolist()
    Tf = g["indices"].shape[2]
    sig_rms = float(np.sqrt((g["recon"].astype(np.float64) ** 2).mean()))
    for b in range(c["batch"]):
        cut = _prefix_before(flips, Tf, m.engine.hop_length, b)
        n = g["recon"].shape[-1] if cut is None else min(cut, g["recon"].shape[-1])
        if n > 0 and not ill:
            bar = 1e-3 if cut is None else max(1e-2, 2.0 / Tf)
            err = rms(r2["recon"][b, :, :n], g["recon"][b, :, :n])
            if cut is not None:
                record_report(name, prefix_samples=n, prefix_rms_error_over_signal_rms=err / sig_rms, bar=bar)
            assert err < bar * sig_rms, (b, n, cut, err / sig_rms)
    tok = torch.from_numpy(g["indices"].astype(np.int64)).permute(1, 2, 0).contiguous()
    w2, emb = m.engine.decode_codes(tok)
    assert rms(emb, g["quantized"]) <= qtol
    w3 = m.engine.decode_emb(torch.from_numpy(g["quantized"]))
    assert w2.shape[-1] == m.engine.decoded_samples(g["indices"].shape[2])
    assert torch.equal(w2, w3) if qtol == 0.0 else rms(w2, w3) < 1e-5 * float(w3.pow(2).mean().sqrt())
    n = g["recon"].shape[-1]
    sc = torch.from_numpy(g["scale"]).view(-1, 1, 1) if "scale" in g else 1.0
    assert rms(w2.cpu()[:, :, :n] * sc, g["recon"]) < WAV_RMS_TOL * float(np.sqrt((g["recon"] ** 2).mean())) * 10


@pytest.mark.parametrize("name", FREQ_ANGLE)
def test_freq_codec_mag_angle_against_reference_golden(name):
    """codec_domain [mag_angle, mag_angle] (conf/freqcodec_mag_angle_16k_n32_600k_step.yaml; codec_freq.py:356-364 encode, :426-434 decode).
    torch.angle of a bin whose imaginary part is rounding noise around a negative real part (the symmetric first STFT frame, DC / Nyquist)
    is +pi or -pi by the FFT's rounding: MANIFEST `angle_conditioning` records that the REFERENCE's own exact (fp64) STFT wraps 34 / 73 bins
    of these fixtures and then emits other codes on 14 of 51 / 12 of 14 frames.  So the fixture cannot pin the codes of ANY second STFT.
    What is pinned instead, each against the real reference's outputs:
      1. the engine's feature tensor equals the reference's modulo 2 pi (angle error weighted by the bin's magnitude), wraps counted;
      2. from the REFERENCE's features (test hook fc_debug_freq_features) the whole path is exact: indices bit for bit, waveform 1e-3 rms;
      3. the decode path from the reference's codes (well-conditioned: softplus, sin * pi, cos / sin) within 1e-3 of the signal's rms;
      4. end to end without the hook, the encoder output stays within 3x the fixture's own conditioning number."""
    from helpers import freq_engine_for
    c = MAN["cases"][name]
    m = freq_engine_for(c["config"], c["weight_seed"])
    assert m.arch.input_channels == 2
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"], c.get("channels", 1)).cuda()
    g = golden(name)
    gf = torch.from_numpy(g["features"]).cuda().contiguous()           # [B, 2, F, frames]
    # 1. features modulo 2 pi
    cap = torch.zeros_like(gf)
    m.engine.debug_freq_features(cap, 1)
    r_own = m.engine.encode(wav, c["n_q"], want_enc_out=True)
    torch.cuda.synchronize()
    mag_ref, mag_eng = gf[:, 0].exp(), cap[:, 0].exp()
    top = float(mag_ref.max())
    assert float((mag_eng - mag_ref).abs().max()) < 2e-5 * top
    d = (cap[:, 1] - gf[:, 1]).abs()
    dw = torch.minimum(d, 2 * np.pi - d)                               # angle error modulo 2 pi
    assert float((dw * mag_ref).max()) < 2e-5 * top, "an angle differs by more than a wrap where the bin is not at the rounding floor"
    wrapped = int((d > 3.0).sum())
    # 2. the path behind the STFT, from the reference's own features
    m.engine.debug_freq_features(gf, 2)
    r = m.engine.encode(wav, c["n_q"], want_enc_out=True)
    assert rms(r["enc_out"], g["encoder_out"]) < 2e-5
    ref = g["indices"].astype(np.int64)
    rep = index_report(r["codes"], ref)
    if rep["mismatched_indices"]:
        from helpers import freq_state_for
        _assert_flips_are_near_ties(freq_state_for(c["config"], c["weight_seed"])[2]["quantizer.rq.model.embed"], g["encoder_out"], ref,
                                    r["codes"], got_enc=r["enc_out"], max_frames=1)
    else:
        assert rms(r["quantized"], g["quantized"]) == 0.0
    m.engine.debug_freq_features(gf, 2)
    r2 = m.engine.encode_decode(wav, c["n_q"], use_scale=True)
    m.engine.check_status()
    sig = float(np.sqrt((g["recon"].astype(np.float64) ** 2).mean()))
    assert tuple(r2["recon"].shape) == g["recon"].shape
    if rep["mismatched_indices"] == 0:
        assert rms(r2["recon"], g["recon"]) < 1e-3 * sig
    # 3. decode path from the reference's codes
    tok = torch.from_numpy(ref).permute(1, 2, 0).contiguous()
    w2, emb = m.engine.decode_codes(tok)
    assert rms(emb, g["quantized"]) == 0.0
    n = g["recon"].shape[-1]
    sc = torch.from_numpy(g["scale"]).view(-1, 1, 1) if "scale" in g else 1.0
    assert rms(w2.cpu()[:, :, :n] * sc, g["recon"]) < 1e-3 * sig
    # 4. end to end with the engine's own STFT: bounded by the fixture's conditioning (how far the reference's exact STFT moves ITS output)
    noise = float(c["stft_self_noise"])
    own = rms(r_own["enc_out"], g["encoder_out"])
    assert own < 3.0 * noise, (own, noise)
    extra = {}
    if name + "_variants" in MAN["cases"]:
        # COMMITTED FACT (oracle/make_golden.py): the real reference on this recording with its STFT evaluated exactly (fp64) -- its own feature
        # tensor and codes.  The wraps are a property of the DOMAIN on speech too: every angle bin the engine wraps against the fixture must be a
        # bin whose angle is +-pi within rounding (the fixture's |angle| > 3), exactly the class of bins the reference's own exact STFT wraps;
        # thread counts change nothing on the reference's side.
        v = golden(name + "_variants")
        f64 = torch.from_numpy(v["features_stft64"])
        gfc = gf.cpu()
        d64 = (f64[:, 1] - gfc[:, 1]).abs()
        ref_wrapped = d64 > 3.0
        eng_wrapped = (d > 3.0).cpu()
        assert int(ref_wrapped.sum()) == MAN["cases"][name + "_variants"]["summary"]["stft64"]["angle_bins_wrapped"]
        assert bool((gfc[:, 1].abs()[eng_wrapped] > 3.0).all()) and bool((gfc[:, 1].abs()[ref_wrapped] > 3.0).all())
        for tag in ("threads1", "threads3"):
            assert np.array_equal(v["indices_" + tag].astype(np.int64), ref)
        ref64_frames = int((v["indices_stft64"].astype(np.int64) != ref).any(0).sum())
        extra = dict(reference_fp64_stft_wrapped_bins=int(ref_wrapped.sum()), wrapped_by_both=int((ref_wrapped & eng_wrapped).sum()),
                     reference_fp64_stft_frames_with_other_codes=ref64_frames,
                     reference_fp64_stft_encoder_out_rms=MAN["cases"][name + "_variants"]["summary"]["stft64"]["encoder_out_rms_diff"])
    record_report(name, angle_bins=int(d.numel()), angle_bins_wrapped_by_the_engine=wrapped,
                  reference_fp64_stft=c["angle_conditioning"], encoder_out_rms_vs_fixture_own_stft=own, stft_self_noise=noise,
                  frames_with_other_codes_own_stft=int((r_own["codes"].cpu().numpy() != ref).any(0).sum()),
                  indices_identical_from_reference_features=rep["mismatched_indices"] == 0, **extra)


@pytest.mark.parametrize("name", [n for n, c in MAN["cases"].items() if c.get("kind") == "freqseg"])
def test_freq_codec_segmented_mode_against_reference_golden(name):
    """FreqCodec._encode / _decode with model_conf.segment_dur (codec_freq.py:303-328,390-404): per-frame STFT codec + the triangle
    overlap-add, against the real reference run in that mode."""
    from helpers import freq_engine_for
    c = MAN["cases"][name]
    m = freq_engine_for(c["config"], c["weight_seed"])
    assert m.arch.segment_length == 2400 and m.arch.segment_stride == 2160
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"], c.get("channels", 1))
    g = golden(name)
    r = m.inference(wav.unsqueeze(1), bit_width=None, use_scale=True)
    m.engine.check_status()
    assert len(r["code_indices"]) == len(c["frames"])
    for f, idx in enumerate(r["code_indices"]):
        assert idx.shape == (c["n_q"], c["batch"], c["frames"][f])
        assert index_report(idx, g[f"indices_{f}"].astype(np.int64))["mismatched_indices"] == 0, f
        assert np.allclose(r["code_embeddings"][f][1].cpu().numpy(), g[f"scale_{f}"], rtol=1e-5)
    assert tuple(r["recon_speech"].shape) == g["recon"].shape
    assert rms(r["recon_speech"], g["recon"]) < 1e-3 * float(np.sqrt((g["recon"] ** 2).mean()))


@pytest.mark.parametrize("cfg_name,seed,B,T,kind,bw,use_scale", [
    ("tinyfreq", 11, 3, 1777, "tones", None, True),        # odd length, 12 STFT frames (even -> recon shorter than the input)
    ("tinyfreq", 11, 1, 400, "noise", 2000, False),        # 3 STFT frames, 2 code frames; reduced bit width; no rescale
    ("tinyfreq640", 12, 2, 5000, "noise", None, True),     # time ratios 2,1,2,1
    ("freqmp", 2, 2, 8000, "tones", 4000, True),           # the recipe shape
    ("freqmpgr8", 6, 1, 6000, "noise", None, True),        # grouped Conv2d / ConvTranspose2d (conv_group_ratio 8)
    ("tinyfreqwn", 14, 3, 1500, "noise", None, True),      # weight_norm 2-D nets (no GroupNorm)
    ("tinyfreqwnc", 15, 2, 3333, "tones", 2000, False),    # weight_norm + causal (time padding on the left, time trim on the right)
    ("tinyfreq640wnc", 16, 2, 4100, "noise", None, True),  # the same with time ratios 2,1,2,1
    ("tinyfreqgr1wnc", 17, 1, 2500, "tones", None, True),  # and with grouped convs (direct kernels without statistics)
])
def test_freq_codec_against_oracle_fresh_inputs(cfg_name, seed, B, T, kind, bw, use_scale):
    from helpers import freq_engine_for, freq_oracle_for
    m, orc = freq_engine_for(cfg_name, seed), freq_oracle_for(cfg_name, seed)
    wav = audio(B, T, 2000 + T, kind)
    o = orc.inference(wav, bit_width=bw, use_scale=use_scale)
    ret = m.inference(wav.cuda().unsqueeze(1), bit_width=bw, use_scale=use_scale)
    m.engine.check_status()
    rep = index_report(ret["code_indices"][0], o["code_indices"][0])
    assert ret["recon_speech"].shape == o["recon_speech"].shape
    assert (ret["code_embeddings"][0][1] is None) == (not use_scale)
    if rep["frames_bad"]:
        _assert_flips_are_near_ties(orc.embed, o["encoder_out"], o["code_indices"][0], ret["code_indices"][0], max_frames=1)
    else:
        ref_rms = float(o["recon_speech"].double().pow(2).mean().sqrt())
        assert rms(ret["recon_speech"], o["recon_speech"]) < 1e-3 * ref_rms
        assert rms(ret["code_embeddings"][0][0], o["code_embeddings"][0][0]) == 0.0
        assert rms(ret["sub_quants"][0], o["sub_quants"][0]) == 0.0


@pytest.mark.parametrize("seed", [0, 1, 2, 4, 6, 7, 8, 9, 11, 1002, 1006, 1008, 1013, 1027])   # >= 1000: weight_norm / causal nets too
def test_freq_codec_random_architectures_against_oracle(seed):
    """config.py::fuzz_freq_recipe_config (three more seeds have goldens from the real reference): n_fft 64 / 128 / 512, STFT hops 16 .. 160,
    time ratios 1 / 2, grouped and dense convs, 1 or 2 residual blocks, with and without LSTM."""
    from helpers import freq_engine_for, freq_oracle_for
    m, orc = freq_engine_for(f"freqfuzz{seed}", seed), freq_oracle_for(f"freqfuzz{seed}", seed)
    B, T = 1 + seed % 3, 900 + 433 * (seed % 5)
    wav = audio(B, T, 4000 + seed, "tones" if seed % 2 else "noise")
    o = orc.inference(wav, bit_width=None, use_scale=True)
    ret = m.inference(wav.cuda().unsqueeze(1), bit_width=None, use_scale=True)
    m.engine.check_status()
    rep = index_report(ret["code_indices"][0], o["code_indices"][0])
    assert ret["recon_speech"].shape == o["recon_speech"].shape
    if rep["frames_bad"]:
        _assert_flips_are_near_ties(orc.embed, o["encoder_out"], o["code_indices"][0], ret["code_indices"][0], max_frames=1)
    else:
        assert rms(ret["recon_speech"], o["recon_speech"]) < 1e-3 * float(o["recon_speech"].double().pow(2).mean().sqrt())
        assert rms(ret["code_embeddings"][0][0], o["code_embeddings"][0][0]) == 0.0


def test_freq_codec_batch_independence_and_determinism():
    """Each utterance of a batch is its own STFT image / GroupNorm statistics / LSTM state: rows of a batched call equal the
    single-utterance calls bit for bit, twice."""
    from helpers import freq_engine_for
    m = freq_engine_for("freqmp", 2)
    wav = audio(5, 24000, 77, "tones").cuda()
    a = m.engine.encode_decode(wav, 32)
    b = m.engine.encode_decode(wav, 32)
    assert torch.equal(a["codes"], b["codes"]) and torch.equal(a["recon"], b["recon"])
    for i in (0, 4):
        one = m.engine.encode_decode(wav[i:i + 1], 32)
        assert torch.equal(one["codes"][:, 0], a["codes"][:, i]) and torch.equal(one["recon"][0], a["recon"][i])
    # 19 utterances: the H = 512 persistent LSTM then runs TWO batch tiles side by side on their own workgroups and barrier words
    big = torch.cat([wav, audio(14, 24000, 78, "noise").cuda()], 0)
    c = m.engine.encode_decode(big, 32)
    assert torch.equal(c["codes"][:, :5], a["codes"]) and torch.equal(c["recon"][:5], a["recon"])
    for i in (16, 18):
        one = m.engine.encode_decode(big[i:i + 1], 32)
        assert torch.equal(one["codes"][:, 0], c["codes"][:, i]) and torch.equal(one["recon"][0], c["recon"][i])
    m.engine.check_status()


def test_freq_codec_gr1_benchmark_configuration_in_a_32_utterance_call():
    """The configuration bench.py's FreqCodec side measurement times (`freqmpgr1`, weight seed 0, engine calls of 32): the grouped direct
    kernels at 257 frequency rows, the H = 512 persistent LSTM with its two batch-tile groups and the two-row-set quantiser (N = 32 x 151 rows
    > 16 per CU) are only live in such a call.  Rows 0, 15, 16, 31 of the 32-call equal the single-utterance calls bit for bit, and rows 0 / 31
    match the CPU oracle (pinned to the real reference on this very configuration: goldens freqmpgr1_b1_t16000 / freqmpgr1_b2_t48000)."""
    from helpers import freq_engine_for, freq_oracle_for
    m, orc = freq_engine_for("freqmpgr1", 0), freq_oracle_for("freqmpgr1", 0)
    wav = audio(32, 48000, 1234, "noise")
    old = m.engine.micro_batch
    m.engine.micro_batch = 32
    try:
        a = m.engine.encode_decode(wav.cuda(), 32, use_scale=True)
        b = m.engine.encode_decode(wav.cuda(), 32, use_scale=True)
    finally:
        m.engine.micro_batch = old
    m.engine.check_status()
    assert a["codes"].shape == (32, 32, 151)
    assert torch.equal(a["codes"], b["codes"]) and torch.equal(a["recon"], b["recon"])
    for i in (0, 15, 16, 31):
        one = m.engine.encode_decode(wav[i:i + 1].cuda(), 32, use_scale=True)
        assert torch.equal(one["codes"][:, 0], a["codes"][:, i]), i
        assert torch.equal(one["recon"][0], a["recon"][i]) and torch.equal(one["quantized"][0], a["quantized"][i]), i
    rows = [0, 31]
    o = orc.inference(wav[rows], bit_width=None, use_scale=True)
    got = a["codes"][:, rows].cpu()
    rep = index_report(got, o["code_indices"][0])
    if rep["frames_bad"]:
        _assert_flips_are_near_ties(orc.embed, o["encoder_out"], o["code_indices"][0], got, max_frames=1)
    else:
        ref_rms = float(o["recon_speech"].double().pow(2).mean().sqrt())
        assert rms(a["recon"][rows], o["recon_speech"]) < 1e-3 * ref_rms
        assert rms(a["quantized"][rows], o["code_embeddings"][0][0]) == 0.0


def test_freq_codec_speech2token_dropin(tmp_path):
    """The reference's own entry point over a FreqCodec config.yaml + model.pth pair."""
    import yaml
    from helpers import freq_state_for, freq_oracle_for
    from funcodec_amd.bin.codec_inference import Speech2Token
    cfg, arch, sd = freq_state_for("tinyfreq", 11)
    with open(tmp_path / "config.yaml", "wt") as f:
        yaml.safe_dump(cfg, f)
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, tmp_path / "model.pth")
    s2t = Speech2Token(str(tmp_path / "config.yaml"), str(tmp_path / "model.pth"), device="cuda")
    orc = freq_oracle_for("tinyfreq", 11)
    wav = audio(2, 3200, 5, "tones")
    idx, embs, recon, subs = s2t(wav, run_mod="inference")
    o = orc.inference(wav, None, True)
    assert torch.equal(idx[0].cpu(), o["code_indices"][0]) and recon.shape == o["recon_speech"].shape
    assert rms(recon, o["recon_speech"]) < 1e-3 * float(o["recon_speech"].pow(2).mean().sqrt())
    tok = idx[0].permute(1, 2, 0).contiguous()
    _, _, w2, _ = s2t(tok, run_mod="decode")
    _, _, w3, _ = s2t(embs[0][0], run_mod="decode_emb")
    assert torch.equal(w2, w3) and w2.shape[-1] == s2t.model.engine.decoded_samples(tok.shape[1])
    assert s2t.model.quantizer.encoder_hop_length == 320


# ---- the drop-in API ------------------------------------------------------------------------------
def test_speech2token_dropin_api(tmp_path):
    from funcodec_amd.bin.codec_inference import Speech2Token, Token2Speech
    from funcodec_amd.synth import make_checkpoint
    cfg_path, pth_path = make_checkpoint(str(tmp_path), "ds320", 0)
    s2t = Speech2Token(cfg_path, pth_path, device="cuda")
    orc = oracle_for("ds320", 0)
    wav = audio(2, 8000, 99, "tones")
    idx, embs, recon, subs = s2t(wav.numpy(), run_mod="inference")          # numpy in, like the reference allows
    o = orc.inference(wav, None, True)
    assert isinstance(idx, list) and idx[0].shape == (32, 2, 25) and idx[0].dtype == torch.int64
    assert torch.equal(idx[0].cpu(), o["code_indices"][0])
    assert embs[0][0].shape == (2, 25, 128) and embs[0][1].shape == (2, 1)
    assert recon.shape == (2, 1, 8000) and subs[0].shape == (32, 2, 128, 25)
    assert rms(recon, o["recon_speech"]) < WAV_RMS_TOL
    # text2audio_inference.py:157 usage pattern
    codec = s2t(wav[:1].unsqueeze(1), run_mod="encode")[0][0].squeeze(1).transpose(0, 1)
    assert codec.shape == (25, 32)
    idx8, _, recon8, _ = s2t(wav, bit_width=4000, run_mod="inference")       # 500 bps per quantiser -> 8
    assert idx8[0].shape[0] == 8 and torch.equal(idx8[0], idx[0][:8])
    # decode with a bit_width: keeps the first nq quantisers (codec_inference.py:121-125)
    tok = idx[0].permute(1, 2, 0).contiguous()
    _, e2, w2, _ = s2t(tok, bit_width=4000, run_mod="decode")
    wref, _ = orc.decode_codes(tok[:, :, :8].cpu())
    assert rms(w2, wref) < WAV_RMS_TOL
    _, _, w3, _ = s2t(embs[0][0], run_mod="decode_emb")
    assert rms(w3, orc.decode_emb(o["code_embeddings"][0][0])) < WAV_RMS_TOL
    assert s2t.model.quantizer.encoder_hop_length == 320 and s2t.model.quantizer.codebook_size == 1024
    t2s = Token2Speech(speech2token=s2t)
    assert rms(t2s(tok), orc.decode_codes(tok.cpu())[0]) < WAV_RMS_TOL
    # dtype="float16": inputs of that dtype accepted, floating outputs returned in it, fp32 arithmetic in between (same indices)
    s16 = Speech2Token(cfg_path, pth_path, device="cuda", dtype="float16")
    i16, e16, r16, q16 = s16(wav.half(), run_mod="inference")
    i32, _, r32, _ = s2t(wav.half().float(), run_mod="inference")
    assert torch.equal(i16[0], i32[0]) and r16.dtype == torch.float16 and e16[0][0].dtype == torch.float16 and q16[0].dtype == torch.float16
    assert torch.equal(r16, r32.half())
    # use_scale=False: reconstruction stays in the normalised domain (encoding_decoding.sh passes this)
    _, embs_ns, recon_ns, _ = s2t(wav, use_scale=False)
    assert embs_ns[0][1] is None
    assert rms(recon_ns, orc.inference(wav, None, False)["recon_speech"]) < WAV_RMS_TOL


# ---- (c) size-independent properties at BASELINE.json's full size ----------------------------------
@pytest.fixture(scope="module")
def config_b():
    m = engine_for("ds640", 0)
    wav = audio(16, 160000, 1234).cuda()
    r = m.engine.encode_decode(wav, 32, use_scale=True)
    torch.cuda.synchronize()
    return m, wav, r


def test_full_size_outputs_are_sane(config_b):
    m, wav, r = config_b
    assert r["codes"].shape == (32, 16, 250) and r["recon"].shape == (16, 1, 160000)
    assert int(r["codes"].min()) >= 0 and int(r["codes"].max()) < 1024
    assert bool(torch.isfinite(r["recon"]).all()) and bool(torch.isfinite(r["quantized"]).all())
    # quantised = sum of the sub-quantiser outputs, in stage order (ddp_core_vq.py:407-408)
    acc = torch.zeros_like(r["sub_quants"][0])
    for i in range(32):
        acc = acc + r["sub_quants"][i]
    assert torch.equal(acc.permute(0, 2, 1), r["quantized"])
    assert float(r["scale"].min()) > 0.09 and float(r["scale"].max()) < 0.11     # 0.1*N(0,1) input


def test_full_size_determinism_and_batch_independence(config_b):
    m, wav, r = config_b
    r2 = m.engine.encode_decode(wav, 32, use_scale=True)
    assert torch.equal(r2["codes"], r["codes"]) and torch.equal(r2["recon"], r["recon"])     # bit-reproducible
    # every op is per-utterance: a sub-batch gives the same bits (the basis of the multi-GPU sharding)
    sub = m.engine.encode_decode(wav[5:8], 32, use_scale=True)
    assert torch.equal(sub["codes"], r["codes"][:, 5:8]) and torch.equal(sub["recon"], r["recon"][5:8])


@pytest.mark.parametrize("kind,cfg,B,T", [("time", "ds640", 3, 24000), ("time", "ds320", 2, 6400), ("time", "tiny", 2, 2500),
                                          ("freq", "freqmpgr1", 2, 16000), ("freq", "freqmpgr1", 3, 160000), ("freq", "freqmp", 1, 16000),
                                          ("freq", "tinyfreqgr1", 3, 2500), ("time", "ds640", 2, 160000)])
def test_results_do_not_depend_on_workspace_contents(kind, cfg, B, T):
    """Every intermediate of a call lives in the caller's workspace (include/funcodec_amd.h), which the engine never clears: a kernel that
    reads a location before this call has written it (a halo row, a padded column, the tail slack the unclamped edge loads of the grouped
    convs may touch) would see whatever the previous call -- or nobody -- left there.  The same call on a workspace filled with 0xFF
    bytes (fp32 NaN, int -1) and with zeros must give the bits of the first run."""
    from helpers import freq_engine_for
    m = (freq_engine_for if kind == "freq" else engine_for)(cfg, 0)
    eng = m.engine
    wav = audio(B, T, 77, "tones").cuda()
    nq = m.arch.num_quantizers
    ref = eng.encode_decode(wav, nq)
    assert eng._ws is not None and torch.isfinite(ref["recon"]).all()
    for fill in (0xFF, 0x00, 0x7F):
        eng._ws.fill_(fill)
        torch.cuda.synchronize()
        out = eng.encode_decode(wav, nq)
        eng.check_status()
        assert torch.equal(out["codes"], ref["codes"]), (cfg, fill)
        assert torch.equal(out["recon"], ref["recon"]), (cfg, fill)
    tok = ref["codes"].permute(1, 2, 0).contiguous()
    w0, _ = eng.decode_codes(tok)
    eng._ws.fill_(0xFF)
    w1, _ = eng.decode_codes(tok)
    assert torch.equal(w0, w1)


def test_full_size_encode_decode_round_trip_and_prefix(config_b):
    m, wav, r = config_b
    tok = r["codes"].permute(1, 2, 0).contiguous()
    w2, emb = m.engine.decode_codes(tok)                    # decode path has no scale: compare un-scaled
    r_ns = m.engine.encode_decode(wav, 32, use_scale=False)
    assert torch.equal(emb, r["quantized"])
    assert torch.equal(w2, r_ns["recon"])
    # fewer quantisers = a prefix of the code stack (residual structure)
    r8 = m.engine.encode(wav, 8)
    assert torch.equal(r8["codes"], r["codes"][:8])
    # first-stage indices really are nearest neighbours of the encoder output (fp64 check on a sample)
    enc = m.engine.encode(wav[:2], 1, want_enc_out=True)
    cfg, arch, sd = state_for("ds640", 0)
    e0 = torch.from_numpy(sd["quantizer.rq.model.embed"][0]).double()
    x = enc["enc_out"].reshape(-1, 128).double().cpu()
    d = (x.pow(2).sum(1, keepdim=True) - 2 * x @ e0.t() + e0.pow(2).sum(1)[None])
    best = d.min(1).values
    chosen = d.gather(1, enc["codes"][0].reshape(-1, 1).cpu()).squeeze(1)
    assert float((chosen - best).max()) < 1e-3


def _check_against(embed, got, ref_idx, ref_enc, ref_recon, hop):
    """Indices bit-exact, else every differing frame must be a PROVEN fp32 tie (at most 1 frame in 250); the waveform is
    checked regardless (whole utterances without a tie, up to the tie otherwise).  Returns the tie proofs."""
    rep = index_report(got["codes"], ref_idx)
    proofs = []
    if rep["mismatched_indices"]:
        proofs = _assert_flips_are_near_ties(embed, ref_enc, ref_idx, got["codes"], got_enc=got.get("enc_out"),
                                             max_frames=max(1, rep["frames"] // 250))
    Tf = ref_idx.shape[2]
    ref_recon = torch.as_tensor(ref_recon)
    for b in range(ref_idx.shape[1]):
        cut = _prefix_before([p[1] for p in proofs], Tf, hop, b)
        n = ref_recon.shape[-1] if cut is None else min(cut, ref_recon.shape[-1])
        if n > 0:
            assert rms(got["recon"][b, :, :n], ref_recon[b, :, :n]) < WAV_RMS_TOL, (b, n)
    return proofs


def test_full_size_matches_the_reference_golden_at_the_benchmark_shape(config_b):
    """BASELINE.json configs[1] itself: utterances 0 and 1 of bench.py's batch (16 x 10 s, seed 1234) against the REAL
    reference's output for exactly those inputs (tests/golden/ds640_b2_t160000.npz, oracle/make_golden.py).
    Measured: 15 969 of the 16 000 indices identical; ONE of the 500 frames (utterance 1, frame 10) differs from stage 1 on,
    where the reference's own margin between the two codes is 3.05e-5 at |dist| = 217 (2 ulp of fp32): a tie the reference
    itself resolves differently with another thread count.  The proof below is what makes that statement checkable."""
    m, wav, r = config_b
    c = MAN["cases"]["ds640_b2_t160000"]
    assert c["samples"] == 160000 and c["audio_seed"] == 1234 and c["config"] == "ds640" and c["weight_seed"] == 0
    assert torch.equal(audio(2, 160000, 1234), wav[:2].cpu())          # the fixture's input IS the head of the benchmark batch
    g = golden("ds640_b2_t160000")
    cfg, arch, sd = state_for("ds640", 0)
    enc = m.engine.encode(wav[:2], 32, want_enc_out=True)
    assert torch.equal(enc["codes"], r["codes"][:, :2])
    got = dict(codes=r["codes"][:, :2], recon=r["recon"][:2], enc_out=enc["enc_out"])
    # COMMITTED FACT instead of an argument: the same two utterances through the REAL reference with one thread, and utterance 1
    # alone (tests/golden/ds640_b2_t160000_variants.npz).  Frames whose codes differ between those runs of the reference itself
    # are ties it resolves differently; the engine may differ from the 8-thread fixture only on such frames.
    v = golden("ds640_b2_t160000_variants")
    ref = g["indices"].astype(np.int64)                                            # [32, 2, 250]
    ref_disagrees = (v["indices_threads1"].astype(np.int64) != ref).any(0)         # [2, 250]
    for key in ("indices_utt1_alone", "indices_utt1_alone_threads3"):
        ref_disagrees[1] |= (v[key].astype(np.int64)[:, 0] != ref[:, 1]).any(0)
    ours_differs = (r["codes"][:, :2].cpu().numpy() != ref).any(0)
    assert ref_disagrees.sum() >= 1, "the fixture variants are expected to show the reference disagreeing with itself"
    unexplained = ours_differs & ~ref_disagrees
    print(f"benchmark shape: engine differs on frames {np.argwhere(ours_differs).tolist()}, the reference differs from itself on "
          f"{np.argwhere(ref_disagrees).tolist()}")
    proofs = _check_against(sd["quantizer.rq.model.embed"], got, ref, g["encoder_out"], g["recon"], 640)
    assert int(unexplained.sum()) == 0 or proofs, "a differing frame that is neither a reference self-disagreement nor a proven tie"
    print(f"benchmark shape vs reference golden: {len(proofs)} tie frame(s) of 500: {proofs}; unexplained by reference runs: {int(unexplained.sum())}")
    assert rms(enc["enc_out"], g["encoder_out"]) < 2e-5
    assert float(((r["scale"][:2].cpu() - torch.from_numpy(g["scale"])).abs() / torch.from_numpy(g["scale"])).max()) < 1e-5
    if not proofs:
        assert rms(r["quantized"][:2], g["quantized"]) == 0.0


def test_full_size_matches_oracle_on_a_sampled_utterance(config_b):
    """A third utterance of the benchmark batch (not in the fixture) against the oracle run here (~1 s per 10 s of audio)."""
    m, wav, r = config_b
    orc = oracle_for("ds640", 0)
    o = orc.inference(wav[3:4].cpu(), None, True)
    enc = m.engine.encode(wav[3:4], 32, want_enc_out=True)
    got = dict(codes=r["codes"][:, 3:4], recon=r["recon"][3:4], enc_out=enc["enc_out"])
    _check_against(orc.embed, got, o["code_indices"][0], o["encoder_out"], o["recon_speech"], 640)


def test_lstm_persistent_kernel_back_to_back_calls_full_size():
    """The persistent recurrence exchanges hidden states through a write-once history buffer read with plain (cacheable)
    loads; two calls on DIFFERENT inputs through the same workspace at the benchmark shape (B=16, T=250) must both match
    torch (a stale cache line from the first call would surface in the second)."""
    m = engine_for("ds640", 0)
    orc = oracle_for("ds640", 0)
    p = "decoder.model.1.lstm"
    for seed in (1, 2):
        x = torch.randn(16, 1024, 250, generator=torch.Generator().manual_seed(seed))
        got = m.engine.lstm_forward(p, x).cpu()
        ref = orc._slstm(x, p)
        assert (got - ref).abs().max().item() < 2e-5, seed


def test_lstm_launch_wavefront_fallback_matches_persistent_kernel():
    """The persistent recurrence (one launch, grid barrier) and the per-step launch path (FC_LSTM_PERSIST=0, used when
    the workgroups cannot all be co-resident) must agree bit for bit and with torch."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, os, torch, numpy as np\n"
        "sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')\n"
        "from helpers import engine_for, oracle_for\n"
        "m = engine_for('ds640', 0); orc = oracle_for('ds640', 0)\n"
        "x = torch.randn(5, 1024, 11, generator=torch.Generator().manual_seed(6))\n"
        "got = m.engine.lstm_forward('encoder.model.16.lstm', x).cpu()\n"
        "ref = orc._slstm(x, 'encoder.model.16.lstm')\n"
        "assert (got - ref).abs().max().item() < 1e-5\n"
        "np.save(sys.argv[1], got.numpy())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("1", "0"):
        path = os.path.join(root, "gpurun_out", f"_lstm_{flag}.npy")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        env = dict(os.environ, FC_LSTM_PERSIST=flag)
        subprocess.run([sys.executable, "-c", code, path], check=True, cwd=root, env=env, timeout=300)
        outs.append(np.load(path))
    assert np.array_equal(outs[0], outs[1])


def test_micro_batching_and_large_batches_are_bit_identical():
    """B = 37 > micro_batch: results must equal the per-utterance results bit for bit (also covers the LSTM batch-tile
    tail, B % 16 != 0), and a 33-utterance single call exercises the per-step LSTM fallback (persistent kernel: B <= 32)."""
    m = engine_for("ds320", 0)
    wav = audio(37, 6400, 4242, "tones").cuda()
    full = m.engine.encode_decode(wav, 32)
    for i in (0, 15, 16, 36):
        one = m.engine.encode_decode(wav[i:i + 1], 32)
        assert torch.equal(one["codes"], full["codes"][:, i:i + 1]) and torch.equal(one["recon"], full["recon"][i:i + 1])
    old = m.engine.micro_batch
    try:
        m.engine.micro_batch = 64
        big = m.engine.encode_decode(wav[:33], 32)
    finally:
        m.engine.micro_batch = old
    assert torch.equal(big["codes"], full["codes"][:, :33]) and torch.equal(big["recon"], full["recon"][:33])
    tok = full["codes"].permute(1, 2, 0).contiguous()
    w, e = m.engine.decode_codes(tok)
    assert w.shape[0] == 37 and torch.equal(e, full["quantized"])


def test_cli_encoding_decoding_pipeline(tmp_path):
    """encoding_decoding.sh stage 1-3 equivalent: scp of ragged wavs -> codecs.txt + wavs -> decode from codecs.txt.
    Shorter utterances are wrap-padded inside a batch exactly like the reference (quirk 4 in SURVEY.md §8b), so the
    oracle is run on the same padded batch."""
    from funcodec_amd import io as fio
    from funcodec_amd.bin.codec_inference import main
    from funcodec_amd.synth import make_checkpoint
    cfg_path, pth_path = make_checkpoint(str(tmp_path / "model"), "ds320", 0)
    orc = oracle_for("ds320", 0)
    lens = [4000, 6400, 3333]
    wavs = audio(3, 6400, 77, "tones")
    scp = tmp_path / "wav.scp"
    with open(scp, "wt") as f:
        for i, n in enumerate(lens):
            p = str(tmp_path / f"u{i}.wav")
            fio.save_audio(wavs[i:i + 1, :n], p, 16000, rescale=False)
            f.write(f"u{i} {p}\n")
    out = str(tmp_path / "out.1")
    main(["--ngpu", "1", "--gpuid_list", "0", "--output_dir", out, "--batch_size", "2", "--sampling_rate", "16000",
          "--config_file", cfg_path, "--model_file", pth_path, "--bit_width", "8000", "--use_scale", "false",
          "--need_indices", "true", "--run_mod", "inference", "--stat_flops", "true",
          "--data_path_and_name_and_type", f"{scp},speech,sound"])
    lines = fio.read_scp(os.path.join(out, "codecs.txt"))
    assert [k for k, _ in lines] == ["u0", "u1", "u2"]
    # oracle on the same wrap-padded batches (batch 0 = u0,u1 ; batch 1 = u2)
    batches = list(fio.iter_batches([(str(scp), "speech", "sound")], 2))
    got = {k: fio.load_codec_json(v) for k, v in lines}
    for keys, b in batches:
        o = orc.inference(b["speech"], bit_width=8000, use_scale=False)
        for i, k in enumerate(keys):
            n = int(b["speech_lengths"][i])
            cl = -(-n // 320)
            ref = o["code_indices"][0][:, i, :cl].numpy().T
            assert got[k].shape == (cl, 16) and np.array_equal(got[k], ref), k
            y, sr = fio.read_wav(os.path.join(out, k + ".wav"))
            assert sr == 16000 and y.shape[0] == n
            r = o["recon_speech"][i, 0, :n].numpy()
            r = r * min(0.99 / np.abs(r).max(), 1.0)                   # save_audio(rescale=True)
            assert np.abs(y - r).max() < 2.0 / 32768
    # file_sampling_rate != model rate (reference :270-273,318-322,352-356): 8 kHz files are resampled in, the
    # reconstruction is resampled back and written at the file rate; lengths stay in file samples
    scp8 = tmp_path / "wav8k.scp"
    p8 = str(tmp_path / "v0.wav")
    x8 = fio.resample(wavs[0:1, :6400], 16000, 8000)
    fio.save_audio(x8, p8, 8000, rescale=False)
    with open(scp8, "wt") as f:
        f.write(f"v0 {p8}\n")
    out8 = str(tmp_path / "out8.1")
    main(["--ngpu", "1", "--gpuid_list", "0", "--output_dir", out8, "--batch_size", "1", "--sampling_rate", "16000",
          "--file_sampling_rate", "8000", "--config_file", cfg_path, "--model_file", pth_path, "--bit_width", "8000",
          "--use_scale", "false", "--need_indices", "true", "--run_mod", "inference",
          "--data_path_and_name_and_type", f"{scp8},speech,sound"])
    y8, sr8 = fio.read_wav(os.path.join(out8, "v0.wav"))
    assert sr8 == 8000 and y8.shape[0] == x8.shape[1]
    x8q, _ = fio.read_wav(p8)
    o8 = orc.inference(fio.resample(torch.from_numpy(x8q)[None], 8000, 16000), bit_width=8000, use_scale=False)
    r8 = fio.resample(o8["recon_speech"], 16000, 8000)[0, 0, :y8.shape[0]].numpy()
    r8 = r8 * min(0.99 / np.abs(r8).max(), 1.0)
    assert np.abs(y8 - r8).max() < 3.0 / 32768
    # decode stage: codecs.txt -> wavs (run_mod=decode, data type codec_json), ark index dump on the way
    out2 = str(tmp_path / "dec.1")
    main(["--ngpu", "1", "--gpuid_list", "0", "--output_dir", out2, "--batch_size", "1", "--sampling_rate", "16000",
          "--config_file", cfg_path, "--model_file", pth_path, "--bit_width", "8000", "--run_mod", "decode",
          "--data_path_and_name_and_type", f"{os.path.join(out, 'codecs.txt')},speech,codec_json"])
    for k, v in lines:
        y, _ = fio.read_wav(os.path.join(out2, k + ".wav"))
        tok = torch.from_numpy(fio.load_codec_json(v))[None]
        ref = orc.decode_codes(tok)[0][0, 0].numpy()
        ref = ref * min(0.99 / np.abs(ref).max(), 1.0)
        assert y.shape == ref.shape and np.abs(y - ref).max() < 2.0 / 32768


def test_boundary_error_behaviour():
    """The reference raises Python exceptions / asserts at its boundary (assert x.dim() == 3, channels <= 2, codec_basic.py:
    342-344); the C ABI reports a status + fc_last_error() and never corrupts: bad sizes, n_q out of range, a workspace that
    is too small, an engine that was never given its weights, out-of-range code indices on the decode path (clamped)."""
    import ctypes as C
    from funcodec_amd.engine import CodecEngine, EngineError
    from funcodec_amd.config import arch_from_config, recipe_config
    m = engine_for("ds320", 0)
    eng = m.engine
    wav = audio(2, 6400, 3, "tones").cuda()
    with pytest.raises(EngineError, match="n_q out of range"):
        eng.encode_decode(wav, 33)
    with pytest.raises(EngineError, match="n_q out of range|bad argument"):   # an empty codes tensor is a null pointer
        eng.encode_decode(wav, 0)
    with pytest.raises(EngineError, match="bad argument"):
        eng.encode_decode(wav[:, :0], 32)
    with pytest.raises((AssertionError, NotImplementedError)):
        m.inference(torch.zeros(1, 3, 100))                            # > 2 channels (codec_basic.py:344)
    # workspace one byte short of what the engine asks for
    need = eng.lib.fc_engine_workspace_bytes(eng._h, 2, 6400)
    ws = torch.empty(need // 2, dtype=torch.uint8, device="cuda")
    codes = torch.empty((32, 2, 20), dtype=torch.int64, device="cuda")
    recon = torch.empty((2, 1, 6400), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = eng.lib.fc_encode_decode(eng._h, p(wav), 2, 6400, 32, 1, p(codes), None, None, None, p(recon), p(ws), ws.numel(),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"workspace too small" in eng.lib.fc_last_error()
    good = eng.encode_decode(wav, 32)                                  # and the engine is still usable afterwards
    assert torch.isfinite(good["recon"]).all()
    # an engine without weights refuses to run
    raw = CodecEngine(arch_from_config(recipe_config("tiny")), "cuda:0")
    with pytest.raises(EngineError, match="finalize|not finalized|weights"):
        raw.encode_decode(wav, 2)
    # decode of out-of-range indices: clamped like an index into the codebook must be, never an out-of-bounds read
    tok = good["codes"].permute(1, 2, 0).contiguous().clone()
    tok[0, 0, 0] = 99999
    tok[0, 1, 1] = -5
    w, _ = eng.decode_codes(tok)
    assert torch.isfinite(w).all()
    with pytest.raises(EngineError, match="outside"):                  # ... and reported (test_deferred_device_errors_are_reported)
        eng.check_status()
