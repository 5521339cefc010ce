"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/laura_*.npz by running the REAL reference LauraTTS
(funcodec.tasks.text2audio_generation.Text2AudioGenTask.build_model -> LauraGenModel, and the whole
funcodec.bin.text2audio_inference.Text2Audio pipeline, via oracle/ref_shim.py) on CPU in the build container, and pins
oracle/laura_oracle.py against it bit for bit while doing so.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_laura.py [CASE ...]

Weights, text inputs and prompt audio are never stored: they are re-created from (config name, seed) by funcodec_amd.synth (numpy
only).  Generation is run GREEDY (sampling=False: `topk(1)`), the only deterministic mode of LauraGenModel.sampling_ids; every
step's log-probability vector is recorded (teacher forcing for the parity tests).
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402

ref_shim.install()

from funcodec_amd.config import arch_from_config, recipe_config  # noqa: E402
from funcodec_amd.laura_config import laura_recipe_config, laura_spec_from_config  # noqa: E402
from funcodec_amd.synth import (make_laura_state_dict, make_state_dict, synthetic_audio, synthetic_text,  # noqa: E402
                                write_checkpoint)
from laura_oracle import LauraOracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# name, config, weight seed, text seed, text lengths, max_length, continual lengths (None = no prompt tokens)
CASES = [
    ("laura_tiny_b3", "tinylaura", 3, 31, [7, 5, 9], 12, None),
    ("laura_tiny_cont_b2", "tinylaura", 4, 32, [6, 11], 9, [5, 3]),
    ("laura_tinyphn_b2", "tinylauraphn", 5, 33, [8, 4], 10, None),
    ("laura_tinyuni_b2", "tinylaurauni", 6, 34, [5, 10], 10, None),
    # <eos> inside the loop: +1.2 on the decoder bias of group 0's <eos> logit ends utterance 0 at step 6 and utterance 1 at
    # step 8 (decode_codec's break / "remove eos token" branches, laura_model.py:520-521,544-546)
    ("laura_tiny_eos_b2", "tinylaura", 8, 37, [6, 9], 16, None, (0, 1.2)),
    # the recipe itself (egs/LibriTTS/text2speech_laura/conf/text2audio_codec_lm_nq2_uni_rel_pos.yaml, 88 M parameters)
    ("laura_recipe_b2", "laura", 0, 35, [21, 13], 10, None),
    ("laura_recipephn_cont_b2", "lauraphn", 1, 36, [17, 30], 8, [12, 7]),
]
# the whole Text2Audio pipeline over a real-size codec: (name, laura config, laura seed, codec config, codec seed, n text tokens,
# n prompt-text tokens, prompt samples, max_length)
E2E_CASES = [
    ("laura_e2e_tinyphn_ds320", "tinylauraphn", 7, "ds320", 0, 9, 4, 4800, 14),
    ("laura_e2e_recipephn_ds640", "lauraphn", 2, "ds640", 0, 12, 5, 9600, 12),
]


def build_reference_model(cfg, sd):
    from funcodec.tasks.text2audio_generation import Text2AudioGenTask
    parser = Text2AudioGenTask.get_parser()
    args = parser.parse_args([])
    for k, v in json.loads(json.dumps(cfg)).items():
        setattr(args, k, v)
    if not hasattr(args, "token_list") or cfg.get("token_list") is None:
        args.token_list = None
    model = Text2AudioGenTask.build_model(args).eval()
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    # only the model's private training-time quantiser is absent from the synthetic checkpoint (not on the inference path)
    assert all(k.startswith("quantizer.rq.model.") for k in missing), missing
    return model


def one_hot_prob(model, codec):
    return torch.nn.functional.one_hot(torch.clamp(codec, 0, model.codebook_size - 1), model.codebook_size).float()


def run_case(name, cfg_name, wseed, tseed, lens, max_length, cont_lens, eos_bias=None):
    cfg = laura_recipe_config(cfg_name)
    spec = laura_spec_from_config(cfg)
    sd = make_laura_state_dict(cfg, wseed, eos_bias=eos_bias)
    model = build_reference_model(cfg, sd)
    orc = LauraOracle(cfg, sd)
    B = len(lens)
    text_in = synthetic_text(cfg, B, lens, tseed)
    rng = np.random.Generator(np.random.PCG64(tseed + 1000))
    continual = None
    if cont_lens is not None:
        continual = [rng.integers(0, spec.codebook_size, size=(n, spec.predict_nq)).astype(np.int64) for n in cont_lens]
    arrays = {}
    with torch.no_grad():
        if spec.vocab_size > 0:
            ids = torch.from_numpy(text_in)
            emb = model.token_embedding(ids.clamp(min=0)) * (ids >= 0).unsqueeze(-1)
            assert torch.equal(emb, orc.token_embed(ids.clamp(min=0)) * (ids >= 0).unsqueeze(-1))
        else:
            emb = torch.from_numpy(text_in)
        tl = torch.tensor(lens, dtype=torch.int64)
        text_outs, out_lens = model.encode(emb, tl)
        assert torch.equal(out_lens, tl)
        o_outs = orc.encode(emb, lens)
        assert torch.equal(o_outs, text_outs), f"{name}: oracle encode != reference"
        arrays["text_outs"] = text_outs.numpy()
        # record every log-prob vector the reference samples from
        rec = []
        real_score = model.codec_lm.score

        def spy(y, state, x):
            out = real_score(y, state, x)
            rec.append(out[0].clone())
            return out

        model.codec_lm.score = spy
        codecs = []
        for b in range(B):
            rec.clear()
            t_b = text_outs[b:b + 1, : lens[b]]
            cont = continual[b].tolist() if continual is not None else None
            dec = model.decode_codec(t_b, tl[b:b + 1], max_length=max_length, sampling=False, beam_size=1, continual=cont)
            logp = torch.stack(rec)
            o_dec, o_logp = orc.decode_codec(t_b[0], max_length, sampling=False, continual=cont, return_logp=True)
            assert torch.equal(o_dec, dec[0]), f"{name}[{b}]: oracle tokens != reference"
            assert torch.equal(o_logp, logp), f"{name}[{b}]: oracle log-probs != reference"
            # one full-sequence pass reproduces every step's scores up to GEMM blocking
            seq = orc.llm_input(t_b[0], dec[0])
            all_lp = orc.lm_score_all(seq, 1 + lens[b])
            n_cont = 0 if cont is None else len(cont)
            steps = logp.shape[0]
            rows = all_lp[lens[b] + 1 + n_cont: lens[b] + 1 + n_cont + steps]
            err = float((rows - logp[: rows.shape[0]]).abs().max())
            assert err < 2e-5, (name, b, err)
            arrays[f"tokens_{b}"] = dec[0].numpy().astype(np.int16)
            arrays[f"logp_{b}"] = logp.numpy()
            codecs.append(dec[0])
        model.codec_lm.score = real_score
        # fine codec predictor on the batch (one-hot probabilities like syn_audio)
        cl = torch.tensor([c.shape[0] for c in codecs], dtype=torch.int64)
        Tc = int(cl.max())
        codec_pad = torch.zeros(B, Tc, spec.predict_nq, dtype=torch.int64)
        for b, c in enumerate(codecs):
            codec_pad[b, : c.shape[0]] = c
        emb_ref, _ = model.cal_codec_emb(text_outs, tl, one_hot_prob(model, codec_pad), cl)
        o_emb = orc.cal_codec_emb([text_outs[b, : lens[b]] for b in range(B)], codecs)
        for b in range(B):
            assert torch.equal(o_emb[b], emb_ref[b, : cl[b]]), f"{name}[{b}]: oracle codec_emb != reference"
            arrays[f"codec_emb_{b}"] = emb_ref[b, : cl[b]].numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrays)
    meta = dict(config=cfg_name, weight_seed=wseed, text_seed=tseed, text_lengths=lens, max_length=max_length,
                continual_lengths=cont_lens, tokens=[int(c.shape[0]) for c in codecs], eos_bias=eos_bias,
                steps=[int(arrays[f"logp_{b}"].shape[0]) for b in range(B)])
    print(name, meta, flush=True)
    return meta


def run_e2e(name, lcfg_name, lseed, ccfg_name, cseed, n_text, n_prompt, prompt_samples, max_length):
    """The REAL Text2Audio.__call__ (bin/text2audio_inference.py:137-198) in continual (zero-shot prompt) mode, greedy."""
    from funcodec.bin.text2audio_inference import Text2Audio
    from make_golden import reference_config
    lcfg = laura_recipe_config(lcfg_name)
    spec = laura_spec_from_config(lcfg)
    lsd = make_laura_state_dict(lcfg, lseed)
    ccfg = recipe_config(ccfg_name)
    csd = make_state_dict(arch_from_config(ccfg), cseed)
    # a trained LauraTTS carries the codec's first codebooks in quantizer_codebook.embed
    lsd["quantizer_codebook.embed"] = csd["quantizer.rq.model.embed"][: spec.num_quantizers].copy()
    toks = lcfg["token_list"]
    rng = np.random.Generator(np.random.PCG64(lseed + 500))
    text = " ".join(toks[i] for i in rng.integers(2, len(toks), size=n_text))
    prompt_text = " ".join(toks[i] for i in rng.integers(2, len(toks), size=n_prompt))
    prompt_audio = synthetic_audio(1, prompt_samples, lseed + 600, "tones")
    with tempfile.TemporaryDirectory() as tmp:
        # build_model_from_file loads tolerantly (filter_state_dict): the private training-time quantiser keeps its init
        lcfg_path, lpth_path = write_checkpoint(os.path.join(tmp, "laura"), lcfg, lsd)
        ccfg_path, cpth_path = write_checkpoint(os.path.join(tmp, "codec"), reference_config(ccfg), csd)
        t2a = Text2Audio(config_file=lcfg_path, model_file=lpth_path, device="cpu", text_emb_model=None, beam_size=1,
                         sampling=False, continual=True, codec_config_file=ccfg_path, codec_model_file=cpth_path,
                         tokenize_to_phone=False, exclude_prompt=True)
        real_decode = t2a.model.decode_codec
        t2a.model.decode_codec = lambda *a, **k: real_decode(*a, **{**k, "max_length": max_length})
        with torch.no_grad():
            ret, decoded = t2a(text, prompt_text, prompt_audio)
    arrays = dict(gen=ret["gen"].numpy(), gen_only_lm=ret["gen_only_lm"].numpy(), decoded_codec=decoded[0].numpy().astype(np.int16))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrays)
    meta = dict(laura_config=lcfg_name, laura_seed=lseed, codec_config=ccfg_name, codec_seed=cseed, text=text,
                prompt_text=prompt_text, prompt_samples=prompt_samples, prompt_audio_seed=lseed + 600, max_length=max_length,
                decoded_frames=int(decoded.shape[1]), gen_samples=int(ret["gen"].shape[-1]))
    print(name, {k: v for k, v in meta.items() if k not in ("text", "prompt_text")}, flush=True)
    return meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*")
    only = set(ap.parse_args().cases) or None
    torch.manual_seed(0)
    path = os.path.join(GOLD, "MANIFEST_laura.json")
    manifest = json.load(open(path)) if os.path.exists(path) else {"cases": {}, "e2e": {}}
    manifest.update(torch=torch.__version__, threads=torch.get_num_threads())
    def save():
        with open(path, "wt") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)

    if only is None or "keys" in only:
        # names and shapes of the REAL model's state_dict for the recipe: the checkpoint-contract fixture of tests/test_laura.py
        cfg = laura_recipe_config("laura")
        model = build_reference_model(cfg, make_laura_state_dict(cfg, 0))
        with open(os.path.join(GOLD, "state_dict_keys_laura.json"), "wt") as f:
            json.dump({k: list(v.shape) for k, v in model.state_dict().items()}, f, indent=0, sort_keys=True)

    for c in CASES:
        if only is None or c[0] in only:
            manifest["cases"][c[0]] = run_case(*c)
            save()
    for c in E2E_CASES:
        if only is None or c[0] in only:
            manifest["e2e"][c[0]] = run_e2e(*c)
            save()


if __name__ == "__main__":
    main()
