"""LauraTTS generation path (BASELINE.json configs[4]).

CPU tests: oracle/laura_oracle.py against the golden vectors the REAL reference produced (oracle/make_golden_laura.py: LauraGenModel
built by Text2AudioGenTask.build_model, Text2Audio.__call__ end to end), the sampling restatement, config validation, the
checkpoint contract.  GPU tests (`-m gpu`): the HIP engine through the C ABI against the same goldens.

Tolerances (north_star: "within the tolerance stated"): fp32 everywhere on both sides, so the bars are fp32 rounding of a
12-layer network -- log-probabilities within 2e-4 absolute of the reference's at every teacher-forced position, greedy tokens
identical (a differing token must be a proven near-tie of the reference's own scores), text-encoder / codec-embedding outputs
within 1e-4 RMS relative to the signal, waveforms within 1e-4 RMS (BASELINE.json's waveform bar).
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLD, golden, rms

from funcodec_amd.laura_config import laura_recipe_config, laura_spec_from_config
from funcodec_amd.synth import laura_plan, make_laura_state_dict, synthetic_audio, synthetic_text

with open(os.path.join(GOLD, "MANIFEST_laura.json")) as f:
    MAN = json.load(f)
CASES = sorted(MAN["cases"])
SMALL = [n for n in CASES if MAN["cases"][n]["config"].startswith("tiny")]
SAME_BUILD = torch.__version__ == MAN["torch"] and torch.get_num_threads() == MAN["threads"]
LOGP_TOL = 2e-4


def case_inputs(name):
    c = MAN["cases"][name]
    cfg = laura_recipe_config(c["config"])
    eos_bias = tuple(c["eos_bias"]) if c.get("eos_bias") else None
    sd = make_laura_state_dict(cfg, c["weight_seed"], eos_bias=eos_bias)
    text = synthetic_text(cfg, len(c["text_lengths"]), c["text_lengths"], c["text_seed"])
    spec = laura_spec_from_config(cfg)
    continual = None
    if c["continual_lengths"] is not None:
        rng = np.random.Generator(np.random.PCG64(c["text_seed"] + 1000))
        continual = [rng.integers(0, spec.codebook_size, size=(n, spec.predict_nq)).astype(np.int64) for n in c["continual_lengths"]]
    return c, cfg, spec, sd, text, continual


# ================================================================ CPU ================================================================
@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_reference_golden(name):
    from laura_oracle import LauraOracle
    c, cfg, spec, sd, text, continual = case_inputs(name)
    g = golden(name)
    orc = LauraOracle(cfg, sd)
    lens = c["text_lengths"]
    with torch.no_grad():
        emb = torch.from_numpy(text)
        if spec.vocab_size > 0:
            emb = orc.token_embed(emb.clamp(min=0)) * (emb >= 0).unsqueeze(-1)
        outs = orc.encode(emb, lens)
        assert rms(outs, g["text_outs"]) < 1e-6
        codecs = []
        for b in range(len(lens)):
            cont = continual[b].tolist() if continual is not None else None
            toks, logp = orc.decode_codec(outs[b, : lens[b]], c["max_length"], sampling=False, continual=cont, return_logp=True)
            assert float((logp - torch.from_numpy(g[f"logp_{b}"])).abs().max()) < 1e-4
            assert np.array_equal(toks.numpy(), g[f"tokens_{b}"].astype(np.int64))
            if SAME_BUILD:
                assert np.array_equal(logp.numpy(), g[f"logp_{b}"])
            assert toks.shape[0] == c["tokens"][b]
            codecs.append(toks)
        embs = orc.cal_codec_emb([outs[b, : lens[b]] for b in range(len(lens))], codecs)
        for b in range(len(lens)):
            assert rms(embs[b], g[f"codec_emb_{b}"]) < 1e-5


def test_eos_case_really_ends_inside_the_loop():
    c = MAN["cases"]["laura_tiny_eos_b2"]
    assert c["tokens"] == [6, 8] and c["steps"] == [7, 9] and c["max_length"] == 16


def test_sampling_restatement_follows_the_reference_candidate_sets():
    """LauraOracle.sample_from against LauraGenModel.sampling_ids' own candidate construction (laura_model.py:466-499): greedy =
    topk(1); int = the k most probable; float = the shortest prefix of the stable descending sort whose mass reaches p; the drawn
    index is the inverse CDF of the given uniform number over that candidate list."""
    from laura_oracle import LauraOracle
    rng = np.random.Generator(np.random.PCG64(5))
    for trial in range(20):
        scores = torch.from_numpy(rng.standard_normal(1025).astype(np.float32) * 2.0)
        probs = scores.softmax(0)
        assert LauraOracle.sample_from(scores, False, 0.3) == int(scores.argmax())
        topv, topi = probs.topk(7)
        for u in (0.0, 0.31, 0.999999):
            got = LauraOracle.sample_from(scores, 7, u)
            assert got in topi.tolist()
            c = torch.cumsum(topv.double(), 0)
            assert got == int(topi[int((c <= u * float(c[-1])).sum().clamp(max=6))])
        sv, si = probs.sort(descending=True, stable=True)
        n = int((torch.cumsum(sv, 0) < 0.8).sum()) + 1
        for u in (0.0, 0.5, 0.97):
            assert LauraOracle.sample_from(scores, 0.8, u) in si[: n + 1].tolist()
        # full softmax: u sweeps the whole CDF
        c = torch.cumsum(probs.double(), 0)
        for u in (0.0, 0.123, 0.5, 0.9999):
            assert LauraOracle.sample_from(scores, True, u) == int((c <= u * float(c[-1])).sum().clamp(max=1024))


def test_config_validation_refuses_what_the_engine_cannot_reproduce():
    cfg = laura_recipe_config("laura")
    spec = laura_spec_from_config(cfg)
    assert (spec.codec_lm.layers, spec.codec_lm.d_model, spec.codec_lm.heads, spec.codec_lm.ff) == (12, 512, 8, 2048)
    assert spec.lm_vocab == 2050 and spec.pos_emb_type == "split" and spec.bidirectional_inputs
    assert spec.text_encoder.act == "swish" and spec.codec_lm.act == "relu" and spec.codec_lm.embed_relu
    for path, val in [(("text_encoder_conf", "use_cnn_module"), True), (("text_encoder_conf", "macaron_style"), True),
                      (("codec_encoder_conf", "rel_pos_type"), "legacy"), (("model_conf", "codec_lm_conf", "pe_type"), "split"),
                      (("model_conf", "codec_lm_conf", "pos_enc"), "abs_pos"), (("model_conf", "codec_conf", "codebook_size"), 512),
                      (("text_encoder_conf", "input_layer"), "conv2d"), (("model_conf", "pos_emb_type"), "other"),
                      # limits of the kernels, refused with a reason at config time instead of a hipErrorInvalidValue at the first call
                      (("text_encoder_conf", "output_size"), 2048), (("model_conf", "codec_lm_conf", "unit"), 4096)]:
        bad = laura_recipe_config("laura")
        d = bad
        for k in path[:-1]:
            d = d[k]
        d[path[-1]] = val
        with pytest.raises(NotImplementedError):
            laura_spec_from_config(bad)
    bad = laura_recipe_config("laura")
    bad["text_encoder"] = "transformer"
    with pytest.raises(NotImplementedError):
        laura_spec_from_config(bad)


def test_checkpoint_plan_matches_the_real_models_state_dict_keys():
    """tests/golden/state_dict_keys_laura.json = names and shapes of the REAL LauraGenModel's state_dict for the recipe
    (written by oracle/make_golden_laura.py); the engine's contract must be a subset that covers everything but the
    training-time quantiser."""
    path = os.path.join(GOLD, "state_dict_keys_laura.json")
    real = {k: tuple(v) for k, v in json.load(open(path)).items()}
    plan = dict(laura_plan(laura_recipe_config("laura")))
    for k, shape in plan.items():
        assert k in real and real[k] == tuple(shape), k
    left = [k for k in real if k not in plan]
    assert all(k.startswith("quantizer.rq.model.") or k == "quantizer_codebook.codec_index_shift" for k in left), left


def test_library_exports_the_laura_entry_points():
    from funcodec_amd import _lib
    lib = _lib.load()
    for name in _lib.SYMBOLS:
        if name.startswith("fc_laura_"):
            assert hasattr(lib, name)
    hdr = open(os.path.join(os.path.dirname(GOLD), "..", "include", "funcodec_amd.h")).read()
    declared = set(__import__("re").findall(r"\b(fc_laura_[a-z_]+)\s*\(", hdr))
    assert declared == {n for n in _lib.SYMBOLS if n.startswith("fc_laura_")}


def test_text2audio_cli_parser_and_sampling_argument():
    """`--sampling` is the reference's int_or_float_or_bool (25 -> top-k, 0.8 -> nucleus, true / false -> softmax / greedy) and maps
    onto the engine's (mode, k, p); the CLI takes demo.sh's command line."""
    from funcodec_amd.bin.text2audio_inference import get_parser
    from funcodec_amd.laura import sampling_args
    a = get_parser().parse_args(["--config_file", "c.yaml", "--model_file", "m.pth", "--codec_config_file", "cc.yaml", "--codec_model_file",
                                 "cm.pth", "--sampling", "25", "--continual", "2500", "--raw_inputs", "text", "--raw_inputs", "prompt text",
                                 "--raw_inputs", "p.wav", "--output_dir", "out", "--log_level", "warning"])
    assert a.sampling == 25 and isinstance(a.sampling, int) and a.continual == 2500 and a.raw_inputs == ["text", "prompt text", "p.wav"]
    assert sampling_args(a.sampling) == (2, 25, 0.0)
    assert sampling_args(True) == (1, 0, 0.0) and sampling_args(False) == (0, 0, 0.0) and sampling_args(0.8) == (3, 0, 0.8)
    assert get_parser().parse_args(["--sampling", "false"]).sampling is False
    assert get_parser().parse_args(["--sampling", "0.9"]).sampling == 0.9
    with pytest.raises(NotImplementedError):
        sampling_args("nucleus")


def test_laura_engine_refuses_cpu():
    from funcodec_amd.engine import EngineError
    from funcodec_amd.laura import LauraEngine
    spec = laura_spec_from_config(laura_recipe_config("tinylaura"))
    with pytest.raises(EngineError):
        LauraEngine(spec, "cpu")
    if not torch.cuda.is_available():
        eng = LauraEngine(spec, "cuda:0")           # plan only; no device work yet
        want = eng.expected_tensors()
        assert set(want) == {k for k, _ in laura_plan(laura_recipe_config("tinylaura"))}
        with pytest.raises(EngineError):
            eng.load_state_dict(make_laura_state_dict(laura_recipe_config("tinylaura"), 0))     # finalize needs a gfx950 device


# ================================================================ GPU ================================================================
_engines = {}


def laura_engine(name):
    from funcodec_amd.laura import LauraGenMI355X
    c, cfg, spec, sd, _, _ = case_inputs(name)
    key = (c["config"], c["weight_seed"], json.dumps(c.get("eos_bias")))
    if key not in _engines:
        if len(_engines) >= 3:
            _engines.pop(next(iter(_engines)))
        m = LauraGenMI355X(spec, "cuda:0", max_positions=256)
        m.load_state_dict(sd)
        _engines[key] = m
    return _engines[key]


def _pad_tokens(toks, nq):
    n = max(t.shape[0] for t in toks)
    out = np.zeros((len(toks), max(n, 1), nq), np.int64)
    for i, t in enumerate(toks):
        out[i, : t.shape[0]] = t
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["laura_tiny_b3", "laura_recipe_b2"])
def test_every_linear_against_torch_cpu(name):
    """Each Linear of the checkpoint through both device forms (implicit-GEMM conv kernel; decoding-step GEMV) against
    torch.nn.functional.linear on CPU."""
    c, cfg, spec, sd, _, _ = case_inputs(name)
    m = laura_engine(name)
    rng = np.random.Generator(np.random.PCG64(1))
    names = ["text_encoder.embed.0", "text_encoder.encoders.0.feed_forward.w_1", "text_encoder.encoders.1.feed_forward.w_2",
             "text_enc_out_layer", "codec_lm.encoder.embed.0", "codec_lm.encoder.encoders.0.self_attn.linear_out",
             "codec_lm.encoder.encoders.1.feed_forward.w_1", "codec_lm.encoder.encoders.1.feed_forward.w_2", "codec_lm.decoder",
             "codec_encoder.encoders.0.self_attn.linear_out", "codec_encoder_out_layer"]
    for n in names:
        W, b = torch.from_numpy(sd[n + ".weight"]), torch.from_numpy(sd[n + ".bias"])
        for (B, T) in ((2, 37), (1, 5), (3, 130)):
            x = torch.from_numpy(rng.standard_normal((B, T, W.shape[1])).astype(np.float32))
            ref = torch.nn.functional.linear(x, W, b)
            got = m.engine.linear(n, x).cpu()
            assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), (n, B, T)
        if n.startswith("codec_lm"):
            x = torch.from_numpy(rng.standard_normal((1, 7, W.shape[1])).astype(np.float32))
            ref = torch.nn.functional.linear(x, W, b)
            got = m.engine.linear(n, x, step_form=True).cpu()
            assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), (n, "step")
    # the fused q / k / v projection
    p = "codec_lm.encoder.encoders.0.self_attn"
    W = torch.cat([torch.from_numpy(sd[f"{p}.linear_{k}.weight"]) for k in "qkv"])
    b = torch.cat([torch.from_numpy(sd[f"{p}.linear_{k}.bias"]) for k in "qkv"])
    x = torch.from_numpy(rng.standard_normal((2, 9, W.shape[1])).astype(np.float32))
    ref = torch.nn.functional.linear(x, W, b)
    assert float((m.engine.linear(p + ".linear_qkv", x).cpu() - ref).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_engine_against_reference_golden(name):
    """encode -> teacher-forced LM scores (full-sequence form) -> greedy decode_codec with a KV cache (step form, per-step
    log-probabilities) -> codec embedding, every stage against what the REAL reference produced."""
    c, cfg, spec, sd, text, continual = case_inputs(name)
    g = golden(name)
    m = laura_engine(name)
    lens, B, nq, V = c["text_lengths"], len(c["text_lengths"]), spec.predict_nq, spec.lm_vocab
    # 1. LauraGenModel.encode
    outs, _ = m.encode(torch.from_numpy(text), torch.tensor(lens))
    ref_outs = torch.from_numpy(g["text_outs"])
    for b in range(B):
        r = ref_outs[b, : lens[b]]
        assert rms(outs[b, : lens[b]], r) < 1e-4 * max(1.0, float(r.pow(2).mean().sqrt())), (name, b)
    assert float(outs.cpu()[0, lens[0]:].abs().max() if lens[0] < outs.shape[1] else 0.0) == 0.0
    # from here on feed the REFERENCE's text_outs so that every stage is compared on identical inputs
    toks_ref = [g[f"tokens_{b}"].astype(np.int64) for b in range(B)]
    cl = c["continual_lengths"] or [0] * B
    # 2. teacher forcing, full-sequence form: the reference's tokens in, log-probabilities of every step out
    codec = torch.from_numpy(_pad_tokens(toks_ref, nq))
    clen = [t.shape[0] for t in toks_ref]
    lp = m.engine.lm_logprobs(ref_outs, lens, codec, clen).cpu()
    for b in range(B):
        ref_lp = torch.from_numpy(g[f"logp_{b}"])              # step s sampled from position text_len + 1 + cont + s
        p0 = lens[b] + 1 + cl[b]
        n = min(ref_lp.shape[0], lp.shape[1] - p0)
        err = float((lp[b, p0: p0 + n] - ref_lp[:n]).abs().max())
        assert err < LOGP_TOL, (name, b, "full-sequence", err)
    # 3. decode_codec, greedy, KV cache: tokens and the per-step log-probabilities
    cont = None if continual is None else torch.from_numpy(_pad_tokens(continual, nq))
    tokens, out_lens, slp = m.engine.decode_codec(ref_outs, lens, c["max_length"], sampling=False, continual=cont,
                                                  continual_lengths=c["continual_lengths"], return_logp=True)
    tokens, slp = tokens.cpu().numpy(), slp.cpu()
    for b in range(B):
        ref_lp = torch.from_numpy(g[f"logp_{b}"])
        got = tokens[b, : out_lens[b]]
        same = got.shape == toks_ref[b].shape and np.array_equal(got, toks_ref[b])
        # log-probabilities agree up to the first differing token (identical histories up to there)
        nsame = 0
        while nsame < min(len(got), len(toks_ref[b])) - cl[b] and np.array_equal(got[cl[b] + nsame], toks_ref[b][cl[b] + nsame]):
            nsame += 1
        ncmp = min(nsame + 1, ref_lp.shape[0])
        err = float((slp[b, :ncmp] - ref_lp[:ncmp]).abs().max())
        assert err < LOGP_TOL, (name, b, "step form", err)
        if not same:
            # a differing greedy token must be a near-tie of the REFERENCE's own scores at that step (its top-2 margin below
            # the engine's measured score error), never a wrong computation
            s = nsame
            assert s < ref_lp.shape[0], (name, b, "length differs without a differing token")
            k = int(np.argmax(got[cl[b] + s] != toks_ref[b][cl[b] + s])) if s < len(got) - cl[b] else 0
            grp = ref_lp[s].reshape(nq, -1)[k]
            top2 = grp.topk(2)[0]
            assert float(top2[0] - top2[1]) < 2 * LOGP_TOL, (name, b, s, "token differs and is not a tie", float(top2[0] - top2[1]))
    # 3b. teacher forcing through the step form: every step's log-probabilities with the reference's tokens forced
    forced = np.zeros((B, c["max_length"], nq), np.int64)
    for b in range(B):
        t = toks_ref[b][cl[b]:]
        forced[b, : t.shape[0]] = t
    _, _, flp = m.engine.decode_codec(ref_outs, lens, c["max_length"], sampling=False, continual=cont,
                                      continual_lengths=c["continual_lengths"], forced=torch.from_numpy(forced), return_logp=True)
    flp = flp.cpu()
    for b in range(B):
        ref_lp = torch.from_numpy(g[f"logp_{b}"])
        n = min(ref_lp.shape[0], toks_ref[b].shape[0] - cl[b] + 1, c["max_length"])
        err = float((flp[b, :n] - ref_lp[:n]).abs().max())
        assert err < LOGP_TOL, (name, b, "forced step form", err)
    # 4. cal_codec_emb on the reference's tokens
    emb = m.engine.codec_emb(ref_outs, lens, codec, clen).cpu()
    for b in range(B):
        r = torch.from_numpy(g[f"codec_emb_{b}"])
        assert rms(emb[b, : clen[b]], r) < 1e-4 * max(1.0, float(r.pow(2).mean().sqrt())), (name, b)
        if clen[b] < emb.shape[1]:
            assert float(emb[b, clen[b]:].abs().max()) == 0.0


@pytest.mark.gpu
def test_decode_is_batch_independent_and_reproducible():
    """An utterance decoded alone equals the same utterance decoded inside a batch (tokens and per-step scores bit for bit: every
    op of the step form is per column), and a sampled generation is a function of its seed."""
    name = "laura_tiny_b3"
    c, cfg, spec, sd, text, _ = case_inputs(name)
    g = golden(name)
    m = laura_engine(name)
    lens = c["text_lengths"]
    outs = torch.from_numpy(g["text_outs"])
    tok, ol, lp = m.engine.decode_codec(outs, lens, 10, sampling=False, return_logp=True)
    for b in range(len(lens)):
        t1, o1, l1 = m.engine.decode_codec(outs[b: b + 1, : lens[b]], [lens[b]], 10, sampling=False, return_logp=True)
        assert o1[0] == ol[b] and torch.equal(t1[0, : o1[0]], tok[b, : ol[b]])
        assert float((l1[0] - lp[b]).abs().max()) < 1e-5
    a = m.engine.decode_codec(outs, lens, 12, sampling=True, seed=1234)
    b_ = m.engine.decode_codec(outs, lens, 12, sampling=True, seed=1234)
    c_ = m.engine.decode_codec(outs, lens, 12, sampling=True, seed=1235)
    assert torch.equal(a[0], b_[0]) and a[1] == b_[1]
    assert not torch.equal(a[0], c_[0])
    for mode in (5, 0.7):
        t, o = m.engine.decode_codec(outs, lens, 8, sampling=mode, seed=7)
        assert all(v == 8 for v in o) and int(t.max()) < 1024 and int(t.min()) >= 0


_fuzz_models = {}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 5, 7, 23])
def test_random_shapes_against_oracle(seed):
    """tools/fuzz_laura.py's trial (the one-off sweep over 36 seeds was clean): a random batch size (1 .. 16), ragged text lengths, with /
    without ragged audio prompts, 1 .. 23 steps, the three tiny configurations in turn (embedding / phoneme inputs, split / uni
    positions) -- text encoder, greedy KV-cached decoding (tokens identical), per-step log-probabilities, fine predictor."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_laura", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                            "tools", "fuzz_laura.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    name, B, steps, prompt, w = mod.trial(seed, _fuzz_models)
    assert w["enc"] < 1e-4 and w["fine"] < 1e-4 and w["step_logp"] < LOGP_TOL and w["greedy_mismatch_utts"] == 0, (name, B, steps, prompt, w)


@pytest.mark.gpu
def test_limits_fail_loudly_and_leave_the_engine_usable():
    """What does not fit the engine's position tables / batch limit is an EngineError carrying the reason, never a truncated result,
    and the engine answers the next call as before; the smallest inputs (one text token, one step) work."""
    from funcodec_amd.engine import EngineError
    name = "laura_tiny_b3"
    c, cfg, spec, sd, text, _ = case_inputs(name)
    g = golden(name)
    m = laura_engine(name)                      # max_positions = 256
    lens = c["text_lengths"]
    outs = torch.from_numpy(g["text_outs"])
    ref = m.engine.decode_codec(outs, lens, 6, sampling=False)
    with pytest.raises(EngineError, match="max_positions"):
        m.engine.decode_codec(outs, lens, 256, sampling=False)
    with pytest.raises(EngineError, match="max_positions"):
        m.encode(torch.from_numpy(synthetic_text(cfg, 1, [300], 3)), torch.tensor([300]))
    with pytest.raises(EngineError, match=r"1 \.\. 16 per call"):
        m.engine.decode_codec(outs[:1].repeat(17, 1, 1), [lens[0]] * 17, 4, sampling=False)
    again = m.engine.decode_codec(outs, lens, 6, sampling=False)
    assert torch.equal(ref[0], again[0]) and ref[1] == again[1]
    one = torch.from_numpy(synthetic_text(cfg, 1, [1], 5))
    o1, l1 = m.encode(one, torch.tensor([1]))
    assert o1.shape[:2] == (1, 1) and bool(torch.isfinite(o1).all())
    t, ol = m.engine.decode_codec(o1, [1], 1, sampling=False)
    assert ol == [1] and t.shape[1] >= 1


@pytest.mark.gpu
def test_wrong_shapes_are_refused_before_the_c_abi_sees_them():
    """The C ABI takes plain pointers and trusts the shapes: a tensor of the wrong width, a lengths list of the wrong length
    or a codec tensor with the wrong column count must be an EngineError on the host side, never an out-of-bounds read on the device."""
    from funcodec_amd.engine import EngineError
    name = "laura_tiny_b3"
    c, cfg, spec, sd, text, _ = case_inputs(name)
    g = golden(name)
    m = laura_engine(name)
    lens = c["text_lengths"]
    outs = torch.from_numpy(g["text_outs"])
    B, L, D = outs.shape
    nq = spec.predict_nq
    ref = m.engine.decode_codec(outs, lens, 4, sampling=False)
    wide = torch.zeros(B, L, D + 16)
    codec = torch.zeros(B, 5, nq, dtype=torch.int64)
    bad_calls = [
        lambda: m.engine.decode_codec(wide, lens, 4, sampling=False),                                   # text_outs width
        lambda: m.engine.decode_codec(outs, lens[:-1], 4, sampling=False),                              # short lengths list
        lambda: m.engine.decode_codec(outs, [L + 1] * B, 4, sampling=False),                            # length past the tensor
        lambda: m.engine.decode_codec(outs, lens, 4, sampling=False, continual=codec[:, :, :1], continual_lengths=[5] * B),
        lambda: m.engine.decode_codec(outs, lens, 4, sampling=False, continual=codec, continual_lengths=[5] * (B - 1)),
        lambda: m.engine.decode_codec(outs, lens, 4, sampling=False, continual=codec),                  # continual without lengths
        lambda: m.engine.decode_codec(outs, lens, 4, sampling=False, forced=torch.zeros(B, 3, nq, dtype=torch.int64)),
        lambda: m.engine.lm_logprobs(wide, lens),
        lambda: m.engine.lm_logprobs(outs, lens, torch.zeros(B, 5, nq + 1, dtype=torch.int64), [5] * B),
        lambda: m.engine.lm_logprobs(outs, lens, codec, [5] * (B + 1)),
        lambda: m.engine.lm_logprobs(outs, lens, codec, [6] * B),
        lambda: m.engine.codec_emb(wide, lens, codec, [5] * B),
        lambda: m.engine.codec_emb(outs, lens, codec[:, :, :0], [5] * B),
        lambda: m.engine.codec_emb(outs, lens[:1], codec, [5] * B),
        lambda: m.engine.codec_emb(outs, lens, codec, [5] * (B - 1)),
        lambda: m.engine.encode(torch.zeros(B, L, spec.input_size + 1), lens) if spec.vocab_size <= 0 else m.engine.encode(torch.zeros(B, L, 3, dtype=torch.int64), lens),
        lambda: m.engine.encode(torch.from_numpy(text), lens + [1]),
    ]
    for i, call in enumerate(bad_calls):
        with pytest.raises(EngineError):
            call()
    again = m.engine.decode_codec(outs, lens, 4, sampling=False)
    assert torch.equal(ref[0], again[0]) and ref[1] == again[1]


@pytest.mark.gpu
def test_step_form_long_sequences_against_full_sequence_oracle():
    """16 utterances (the step form's maximum), recipe-size LM, 300 teacher-forced steps: the KV-cached step form at key counts where
    the step attention walks SEVERAL 128-key passes per key range (B x H = 128 workgroup slots -> 2 ranges of ~160 keys) and the
    full-sequence attention spans 21 key tiles, against the oracle's one-pass scores of the same sequences."""
    from laura_oracle import LauraOracle
    cfg = laura_recipe_config("lauraphn")
    spec = laura_spec_from_config(cfg)
    sd = make_laura_state_dict(cfg, 1)
    from funcodec_amd.laura import LauraGenMI355X
    m = LauraGenMI355X(spec, "cuda:0", max_positions=512)
    m.load_state_dict(sd)
    orc = LauraOracle(cfg, sd)
    B, steps = 16, 300
    lens = [10 + (3 * i) % 17 for i in range(B)]
    ids = synthetic_text(cfg, B, lens, 41)
    rng = np.random.Generator(np.random.PCG64(9))
    forced = rng.integers(0, spec.codebook_size, size=(B, steps, spec.predict_nq)).astype(np.int64)
    with torch.no_grad():
        outs, _ = m.encode(torch.from_numpy(ids), torch.tensor(lens))
        _, out_lens, slp = m.engine.decode_codec(outs, lens, steps, sampling=False, forced=torch.from_numpy(forced), return_logp=True)
        assert all(v == steps for v in out_lens)
        full = m.engine.lm_logprobs(outs, lens, torch.from_numpy(forced), [steps] * B).cpu()
        slp = slp.cpu()
        for b in range(B):
            p0 = lens[b] + 1
            # the engine's two forms agree with each other at every step ...
            assert float((slp[b] - full[b, p0: p0 + steps]).abs().max()) < 1e-4, b
        for b in (0, 7, 15):         # ... and with the CPU oracle (three utterances: ~1 s each on the host)
            seq = orc.llm_input(outs[b, : lens[b]].cpu(), torch.from_numpy(forced[b]))
            ref = orc.lm_score_all(seq, 1 + lens[b])
            p0 = lens[b] + 1
            assert float((slp[b] - ref[p0: p0 + steps]).abs().max()) < LOGP_TOL, b


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,B,steps", [("lauraphn", 8, 120), ("lauraphn", 16, 40), ("lauraphn", 3, 40), ("tinylaura", 5, 30)])
def test_persistent_step_equals_the_kernel_chain(cfg_name, B, steps):
    """The decoding step as ONE persistent launch (csrc/laura_persist.hip: workgroups hand the token vectors over through arrival
    counters, weights prefetched by LDS DMA) against the chain of one kernel per Linear / attention it replaces: same arithmetic up to the
    split of the contraction / key ranges, so teacher-forced log-probabilities agree to fp32 rounding at EVERY step and greedy generations
    are identical; repeated runs of the persistent form are bit-identical (fixed summation orders, no atomics on data)."""
    from funcodec_amd.laura import LauraGenMI355X
    cfg = laura_recipe_config(cfg_name)
    spec = laura_spec_from_config(cfg)
    sd = make_laura_state_dict(cfg, 3)
    m = LauraGenMI355X(spec, "cuda:0", max_positions=512)
    m.load_state_dict(sd)
    lens = [12 + (5 * i) % 23 for i in range(B)]
    text = synthetic_text(cfg, B, lens, 17)
    rng = np.random.Generator(np.random.PCG64(4))
    forced = torch.from_numpy(rng.integers(0, spec.codebook_size, size=(B, steps, spec.predict_nq)).astype(np.int64))
    cont = torch.from_numpy(rng.integers(0, spec.codebook_size, size=(B, 9, spec.predict_nq)).astype(np.int64))
    cl = [9 - (i % 4) for i in range(B)]
    with torch.no_grad():
        outs, _ = m.encode(torch.from_numpy(text), torch.tensor(lens))
        assert m.engine.set_persistent_step(True), "the persistent step must be available for this model on MI355X"
        tp, lp, sp = m.engine.decode_codec(outs, lens, steps, sampling=False, forced=forced, return_logp=True, continual=cont, continual_lengths=cl)
        tp2, lp2, sp2 = m.engine.decode_codec(outs, lens, steps, sampling=False, forced=forced, return_logp=True, continual=cont, continual_lengths=cl)
        gp = m.engine.decode_codec(outs, lens, steps, sampling=False)
        kp = m.engine.decode_codec(outs, lens, steps, sampling=25, seed=11)
        assert not m.engine.set_persistent_step(False)
        tc, lc, sc = m.engine.decode_codec(outs, lens, steps, sampling=False, forced=forced, return_logp=True, continual=cont, continual_lengths=cl)
        gc = m.engine.decode_codec(outs, lens, steps, sampling=False)
        m.engine.set_persistent_step(True)
    assert lp == lc == [c + steps for c in cl] and torch.equal(tp, tc)
    assert torch.equal(sp, sp2) and torch.equal(tp, tp2)                      # run-to-run bit-identical
    err = float((sp - sc).abs().max())
    assert err < 2e-5, err
    assert gp[1] == gc[1] and torch.equal(gp[0], gc[0])                       # greedy generations identical
    assert all(0 <= v - c <= steps for v, c in zip(kp[1], [0] * B)) and int(kp[0].max()) < spec.codebook_size   # top-k sampling may draw <eos>


@pytest.mark.gpu
def test_persistent_step_is_bit_stable_under_concurrent_load():
    """The in-launch hand-offs of the persistent step (write-through stores + arrival counters, then PLAIN loads of the edge buffers) are
    only correct if no consumer can see a stale line.  Idle chips and uniform load hide such failures (MI355X_MICROARCH.md): run the
    recipe-size step 24 times while ANOTHER stream keeps the memory system and the CUs busy with the codec engine (a different load
    every round), and require every teacher-forced log-probability of every run to be bit-identical to the first, unloaded one.
    (The side load is the SoundStream-shaped codec, which has no LSTM: two PERSISTENT kernels -- this step and the persistent LSTM -- must not
    run concurrently on one device, each needs every one of its workgroups resident; see INTEGRATION.md.)"""
    from funcodec_amd.laura import LauraGenMI355X
    from helpers import engine_for
    cfg = laura_recipe_config("lauraphn")
    spec = laura_spec_from_config(cfg)
    m = LauraGenMI355X(spec, "cuda:0", max_positions=512)
    m.load_state_dict(make_laura_state_dict(cfg, 6))
    B, steps = 8, 48
    lens = [20 + 3 * i for i in range(B)]
    text = synthetic_text(cfg, B, lens, 23)
    rng = np.random.Generator(np.random.PCG64(8))
    forced = torch.from_numpy(rng.integers(0, spec.codebook_size, size=(B, steps, spec.predict_nq)).astype(np.int64))
    codec = engine_for("ss320", 0)
    assert codec.arch.lstm_layers == 0
    wav = torch.from_numpy(synthetic_audio(6, 48000, 5)).cuda()
    side = torch.cuda.Stream()
    main = torch.cuda.Stream()                      # the decode loop replays a graph: not on the legacy default stream
    with torch.no_grad():
        outs, _ = m.encode(torch.from_numpy(text), torch.tensor(lens))
        assert m.engine.set_persistent_step(True)
        with torch.cuda.stream(main):
            _, _, ref = m.engine.decode_codec(outs, lens, steps, sampling=False, forced=forced, return_logp=True)
        torch.cuda.synchronize()
        for it in range(24):
            with torch.cuda.stream(side):
                for _ in range(1 + it % 4):
                    codec.engine.encode_decode(wav[: 1 + it % 6], 32)
            with torch.cuda.stream(main):
                _, _, lp = m.engine.decode_codec(outs, lens, steps, sampling=False, forced=forced, return_logp=True)
            torch.cuda.synchronize()
            assert torch.equal(lp, ref), (it, float((lp - ref).abs().max()))
    codec.engine.check_status()


@pytest.mark.gpu
def test_persistent_step_timeout_is_loud_and_falls_back(monkeypatch):
    """A hand-off of the persistent step that never completes (a workgroup not resident, a lost store): bounded spins, no hang, no silent
    tokens -- THAT call is re-run on the kernel chain and returns the chain's result with a RuntimeWarning, the fallback is counted
    (fc_laura_persistent_step_fallbacks), and the engine stays on the chain until it is switched back."""
    name = "laura_tiny_b3"
    c, cfg, spec, sd, text, _ = case_inputs(name)
    g = golden(name)
    m = laura_engine(name)
    lens = c["text_lengths"]
    outs = torch.from_numpy(g["text_outs"])
    assert m.engine.set_persistent_step(True)
    ref = m.engine.decode_codec(outs, lens, 8, sampling=False)
    before = m.engine.persistent_step_fallbacks
    monkeypatch.setenv("FC_LAURA_PERSIST_TEST", "timeout")
    with pytest.warns(RuntimeWarning, match="timed out at a hand-off"):
        fell = m.engine.decode_codec(outs, lens, 8, sampling=False)
    monkeypatch.delenv("FC_LAURA_PERSIST_TEST")
    assert fell[1] == ref[1] and torch.equal(fell[0], ref[0])              # the chain's result of the SAME call
    assert m.engine.persistent_step_fallbacks == before + 1
    again = m.engine.decode_codec(outs, lens, 8, sampling=False)           # still on the kernel chain, no new fallback
    assert again[1] == ref[1] and torch.equal(again[0], ref[0]) and m.engine.persistent_step_fallbacks == before + 1
    assert m.engine.set_persistent_step(True)                              # and back
    back = m.engine.decode_codec(outs, lens, 8, sampling=False)
    assert back[1] == ref[1] and torch.equal(back[0], ref[0])


@pytest.mark.gpu
def test_top_k_with_hundreds_of_tied_logits_is_deterministic():
    """A constant output layer makes all 1025 logits of a group equal: more than the 256 slots of the radix-select path tie at the k-th
    key.  The sampler must then order the candidates by (probability descending, index ascending) -- i.e. the top-k set is ids 0 .. k-1 --
    instead of whatever 256 entries its atomics happened to collect, and a generation stays a function of its seed."""
    from funcodec_amd.laura import LauraGenMI355X
    cfg = laura_recipe_config("tinylaura")
    spec = laura_spec_from_config(cfg)
    sd = make_laura_state_dict(cfg, 5)
    sd["codec_lm.decoder.weight"] = np.zeros_like(sd["codec_lm.decoder.weight"])
    sd["codec_lm.decoder.bias"] = np.zeros_like(sd["codec_lm.decoder.bias"])
    m = LauraGenMI355X(spec, "cuda:0", max_positions=256)
    m.load_state_dict(sd)
    lens = [9, 7, 11]
    text = synthetic_text(cfg, 3, lens, 2)
    with torch.no_grad():
        outs, _ = m.encode(torch.from_numpy(text), torch.tensor(lens))
        for persist in (True, False):
            m.engine.set_persistent_step(persist)
            a = m.engine.decode_codec(outs, lens, 20, sampling=5, seed=77)
            b = m.engine.decode_codec(outs, lens, 20, sampling=5, seed=77)
            assert torch.equal(a[0], b[0]) and a[1] == b[1] == [20, 20, 20]
            assert int(a[0].max()) < 5 and int(a[0].min()) >= 0, a[0].unique()
            assert len(a[0].unique()) > 1                     # it does sample among the five
        m.engine.set_persistent_step(True)


@pytest.mark.gpu
def test_device_sampler_follows_the_step_distribution():
    """Device-side sampling against the oracle's restatement of LauraGenModel.sampling_ids on the SAME scores: for top-k the drawn
    ids must lie in the reference's candidate set, and over many seeds the empirical distribution of the first sampled token
    matches softmax of the first step's scores (chi-square over the 8 most probable ids + rest)."""
    name = "laura_tiny_b3"
    c, cfg, spec, sd, text, _ = case_inputs(name)
    g = golden(name)
    m = laura_engine(name)
    lens = c["text_lengths"]
    outs = torch.from_numpy(g["text_outs"])[:1, : lens[0]]
    ref_lp = torch.from_numpy(g["logp_0"])[0].reshape(spec.predict_nq, -1)
    p0 = ref_lp[0].softmax(0)
    top8 = p0.topk(8)[1]
    N = 400
    counts = np.zeros(9)
    top5 = set(p0.topk(5)[1].tolist())
    top100 = set(p0.topk(100)[1].tolist())
    sv, si = p0.sort(descending=True, stable=True)
    nucleus = set(si[: int((torch.cumsum(sv, 0) < 0.7).sum()) + 2].tolist())
    seen100, seen_n = set(), set()
    for s in range(N):
        t, _ = m.engine.decode_codec(outs, [lens[0]], 1, sampling=True, seed=1000 + s)
        tid = int(t[0, 0, 0])
        hit = (top8 == tid).nonzero()
        counts[int(hit[0]) if len(hit) else 8] += 1
        if s < 60:
            t5, _ = m.engine.decode_codec(outs, [lens[0]], 1, sampling=5, seed=s)
            assert int(t5[0, 0, 0]) in top5
            # k > 64 takes the sampler's sort path instead of the radix select; a float is nucleus sampling (the shortest prefix
            # of the descending sort whose mass reaches p; one more candidate allowed for a cumulative sum rounding across p)
            t100, _ = m.engine.decode_codec(outs, [lens[0]], 1, sampling=100, seed=s)
            assert int(t100[0, 0, 0]) in top100
            tn, _ = m.engine.decode_codec(outs, [lens[0]], 1, sampling=0.7, seed=s)
            assert int(tn[0, 0, 0]) in nucleus
            seen100.add(int(t100[0, 0, 0])); seen_n.add(int(tn[0, 0, 0]))
    assert len(seen100) > 5 and len(seen_n) > 1, (seen100, seen_n)      # they do sample, not return the arg-max
    expect = np.concatenate([p0[top8].numpy(), [1.0 - float(p0[top8].sum())]]) * N
    chi2 = float(((counts - expect) ** 2 / np.maximum(expect, 1e-9)).sum())
    assert chi2 < 27.9, (chi2, counts, expect)        # chi-square, 8 degrees of freedom, p = 0.0005


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MAN["e2e"]))
def test_text2audio_dropin_against_the_reference_pipeline(name, tmp_path):
    """The drop-in Text2Audio (config.yaml + model.pth of both models on disk, same constructor keywords, same call) against the
    REAL funcodec.bin.text2audio_inference.Text2Audio run greedily in zero-shot (continual) mode: prompt audio -> codec tokens,
    text -> tokens -> embeddings -> waveforms."""
    from funcodec_amd.bin.text2audio_inference import Text2Audio
    from funcodec_amd.config import arch_from_config, recipe_config
    from funcodec_amd.synth import make_state_dict, write_checkpoint
    c = MAN["e2e"][name]
    g = golden(name)
    lcfg = laura_recipe_config(c["laura_config"])
    spec = laura_spec_from_config(lcfg)
    lsd = make_laura_state_dict(lcfg, c["laura_seed"])
    ccfg = recipe_config(c["codec_config"])
    csd = make_state_dict(arch_from_config(ccfg), c["codec_seed"])
    lsd["quantizer_codebook.embed"] = csd["quantizer.rq.model.embed"][: spec.num_quantizers].copy()
    lc, lp = write_checkpoint(str(tmp_path / "laura"), lcfg, lsd)
    cc, cp = write_checkpoint(str(tmp_path / "codec"), ccfg, csd)
    t2a = Text2Audio(config_file=lc, model_file=lp, device="cuda", text_emb_model=None, beam_size=1, sampling=False, continual=True,
                     codec_config_file=cc, codec_model_file=cp, tokenize_to_phone=False, exclude_prompt=True,
                     max_length=c["max_length"], max_positions=256)
    prompt_audio = synthetic_audio(1, c["prompt_samples"], c["prompt_audio_seed"], "tones")
    ret, decoded = t2a(c["text"], c["prompt_text"], prompt_audio)
    ref_codec = g["decoded_codec"].astype(np.int64)
    got = decoded[0].cpu().numpy()
    assert got.shape == ref_codec.shape, (got.shape, ref_codec.shape)
    n_prompt = ref_codec.shape[0] - c["max_length"]
    assert np.array_equal(got[:n_prompt], ref_codec[:n_prompt]), "prompt audio -> codec tokens differ"
    assert np.array_equal(got, ref_codec), "generated tokens differ"
    for key in ("gen", "gen_only_lm"):
        ref = torch.from_numpy(g[key])
        assert tuple(ret[key].shape) == tuple(ref.shape), (key, ret[key].shape, ref.shape)
        assert rms(ret[key], ref) < 1e-4, (key, rms(ret[key], ref), float(ref.pow(2).mean().sqrt()))
