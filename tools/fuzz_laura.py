"""One-off sweep (not a test): random batch sizes / ragged text lengths / prompt (continual) lengths / step counts through the LauraTTS engine
against the CPU oracle on the tiny configurations: text encoder, KV-cached greedy decoding (tokens must be identical unless the oracle's own
top-2 scores are within 1e-5: reported, never seen so far), per-step log-probabilities, teacher-forced scores and the fine predictor.
usage: python tools/fuzz_laura.py [n_trials] [first_seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from funcodec_amd.laura import LauraGenMI355X  # noqa: E402
from funcodec_amd.laura_config import laura_recipe_config, laura_spec_from_config  # noqa: E402
from funcodec_amd.synth import make_laura_state_dict, synthetic_text  # noqa: E402
from laura_oracle import LauraOracle  # noqa: E402


def trial(seed, models):
    rng = np.random.Generator(np.random.PCG64(seed))
    name = ["tinylaura", "tinylauraphn", "tinylaurauni"][seed % 3]
    if name not in models:
        cfg = laura_recipe_config(name)
        spec = laura_spec_from_config(cfg)
        sd = make_laura_state_dict(cfg, 11)
        m = LauraGenMI355X(spec, "cuda:0", max_positions=192)
        m.load_state_dict(sd)
        models[name] = (cfg, spec, m, LauraOracle(cfg, sd))
    cfg, spec, m, orc = models[name]
    B = int(rng.integers(1, 17))
    lens = [int(v) for v in rng.integers(1, 40, size=B)]
    steps = int(rng.integers(1, 24))
    nq = spec.predict_nq
    text = torch.from_numpy(synthetic_text(cfg, B, lens, 1000 + seed))
    worst = {}
    with torch.no_grad():
        # phoneme ids (padded with -1): model.token_embedding first (Text2Audio.tokenize_text), padding rows zero
        ref_in = text if text.is_floating_point() else orc.token_embed(text.clamp(min=0)) * (text >= 0).unsqueeze(-1)
        ref_outs = orc.encode(ref_in, lens)
        outs, _ = m.encode(text, torch.tensor(lens))
        worst["enc"] = max(float((outs[b, :lens[b]].cpu() - ref_outs[b, :lens[b]]).abs().max()) for b in range(B))
        use_prompt = bool(rng.integers(0, 2))
        cl = [int(v) for v in rng.integers(1, 12, size=B)] if use_prompt else None
        cont = None
        if use_prompt:
            cont = torch.from_numpy(rng.integers(0, spec.codebook_size, size=(B, max(cl), nq)).astype(np.int64))
        tok, ol, lp = m.engine.decode_codec(ref_outs, lens, steps, sampling=False, seed=0, continual=cont, continual_lengths=cl, return_logp=True)
        bad, e_lp = 0, 0.0
        for b in range(B):
            c = None if cont is None else cont[b, :cl[b]].tolist()
            rt, rl = orc.decode_codec(ref_outs[b, :lens[b]], steps, sampling=False, continual=c, return_logp=True)
            got = tok[b, :ol[b]].cpu()
            if ol[b] != rt.shape[0] or not bool((got == rt).all()):
                bad += 1
            e_lp = max(e_lp, float((lp[b, :rl.shape[0]].cpu() - rl).abs().max()))
        worst["greedy_mismatch_utts"] = bad
        worst["step_logp"] = e_lp
        # fine predictor on random codes
        Tc = [int(v) for v in rng.integers(1, 30, size=B)]
        codes = torch.from_numpy(rng.integers(0, spec.codebook_size, size=(B, max(Tc), nq)).astype(np.int64))
        ref_emb = orc.cal_codec_emb([ref_outs[b, :lens[b]] for b in range(B)], [codes[b, :Tc[b]] for b in range(B)])
        emb = m.engine.codec_emb(ref_outs, lens, codes, Tc)
        worst["fine"] = max(float((emb[b, :Tc[b]].cpu() - ref_emb[b]).abs().max()) for b in range(B))
    m.engine.check_status() if hasattr(m.engine, "check_status") else None
    return name, B, steps, use_prompt, worst


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    models, fails = {}, 0
    for seed in range(s0, s0 + n):
        name, B, steps, prompt, w = trial(seed, models)
        ok = w["enc"] < 1e-4 and w["greedy_mismatch_utts"] == 0 and w["step_logp"] < 2e-4 and w["fine"] < 1e-4
        fails += 0 if ok else 1
        print(f"seed {seed:3d} {name:13s} B={B:2d} steps={steps:2d} prompt={int(prompt)}: enc {w['enc']:.1e} logp {w['step_logp']:.1e} fine {w['fine']:.1e} "
              f"greedy mismatches {w['greedy_mismatch_utts']} {'ok' if ok else 'FAIL'}", flush=True)
    print(f"{n} trials, {fails} failures")


if __name__ == "__main__":
    main()
