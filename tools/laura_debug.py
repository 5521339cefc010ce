"""GPU debugging aid for the LauraTTS engine: walks the path stage by stage against the CPU oracle and PRINTS the error of every
stage (never asserts), including the intermediate tensors of the full-sequence stacks through fc_laura_debug_probe, so that one
run on the GPU box localises a wrong kernel.

    python tools/laura_debug.py [tinylaura|laura ...]
"""
import ctypes as C
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from funcodec_amd import _lib  # noqa: E402
from funcodec_amd.laura import LauraGenMI355X  # noqa: E402
from funcodec_amd.laura_config import laura_recipe_config, laura_spec_from_config  # noqa: E402
from funcodec_amd.synth import make_laura_state_dict, synthetic_text  # noqa: E402
from laura_oracle import LauraOracle, make_pad_mask  # noqa: E402


def err(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return f"max|d| {float((a - b).abs().max()):.3e}  rms {float((a - b).pow(2).mean().sqrt()):.3e}  (ref rms {float(b.pow(2).mean().sqrt()):.3e}, nan {int(torch.isnan(a).sum())})"


def probe_stack(m, orc_stack, stack_id, run, x_in, mask, lens, d_rows):
    """run(): launches the engine call that executes the stack; compares every probe point."""
    lib = _lib.load()
    tr = orc_stack.trace(x_in, mask)
    B, T = x_in.shape[0], x_in.shape[1]
    Tp = (T + 3) & ~3
    for (what, layer), ref in sorted(tr.items(), key=lambda kv: (kv[0][1], {5: 0, 0: 0, 1: 1, 2: 2, 3: 3}[kv[0][0]])):
        rows = ref.shape[-1]
        buf = torch.zeros(B * rows * Tp, dtype=torch.float32, device="cuda")
        lib.fc_laura_debug_probe(C.c_void_p(buf.data_ptr()), buf.numel() * 4, stack_id, layer, what)
        run()
        torch.cuda.synchronize()
        lib.fc_laura_debug_probe(None, 0, -1, -1, -1)
        got = buf.view(B, rows, Tp).permute(0, 2, 1)[:, :T].cpu()
        parts = []
        for b in range(B):
            parts.append(err(got[b, : lens[b]], ref[b, : lens[b]]))
        name = {0: "embed out", 1: "LN1 out", 2: "qkv", 3: "ctx", 5: "x at entry"}[what]
        print(f"   stack {stack_id} block {layer} {name:10s}: " + " | ".join(parts), flush=True)


def main():
    names = sys.argv[1:] or ["tinylaura"]
    for cfg_name in names:
        print("=" * 20, cfg_name, flush=True)
        cfg = laura_recipe_config(cfg_name)
        spec = laura_spec_from_config(cfg)
        sd = make_laura_state_dict(cfg, 3)
        orc = LauraOracle(cfg, sd)
        m = LauraGenMI355X(spec, "cuda:0", max_positions=256)
        m.load_state_dict(sd)
        print("engine finalized", flush=True)
        lens = [7, 5, 9]
        B = len(lens)
        text = synthetic_text(cfg, B, lens, 31)
        emb = torch.from_numpy(text)
        if spec.vocab_size > 0:
            emb = orc.token_embed(emb.clamp(min=0)) * (emb >= 0).unsqueeze(-1)
        with torch.no_grad():
            # ---- linears
            for n in ["text_encoder.embed.0", "text_enc_out_layer", "codec_lm.encoder.encoders.0.feed_forward.w_1",
                      "codec_lm.encoder.encoders.0.feed_forward.w_2", "codec_lm.decoder"]:
                try:
                    W, bb = torch.from_numpy(sd[n + ".weight"]), torch.from_numpy(sd[n + ".bias"])
                    x = torch.randn(2, 11, W.shape[1])
                    ref = torch.nn.functional.linear(x, W, bb)
                    print(f" linear {n:48s} full: {err(m.engine.linear(n, x), ref)}", flush=True)
                    if n.startswith("codec_lm"):
                        x = torch.randn(1, 5, W.shape[1])
                        ref = torch.nn.functional.linear(x, W, bb)
                        print(f" linear {n:48s} step: {err(m.engine.linear(n, x, step_form=True), ref)}", flush=True)
                except Exception:
                    traceback.print_exc()
            # ---- text encoder
            ref_outs = orc.encode(emb, lens)
            try:
                outs, _ = m.encode(torch.from_numpy(text), torch.tensor(lens))
                for b in range(B):
                    print(f" encode[{b}]: {err(outs[b, :lens[b]], ref_outs[b, :lens[b]])}", flush=True)
                mask = (~make_pad_mask(lens, emb.size(1)))[:, None, :]
                probe_stack(m, orc.text_encoder, 0, lambda: m.encode(torch.from_numpy(text), torch.tensor(lens)), emb.float(), mask, lens, None)
            except Exception:
                traceback.print_exc()
            # ---- LM teacher forcing
            try:
                toks = [orc.decode_codec(ref_outs[b, : lens[b]], 6) for b in range(B)]
                Tc = max(t.shape[0] for t in toks)
                codec = torch.zeros(B, Tc, spec.predict_nq, dtype=torch.int64)
                for b, t in enumerate(toks):
                    codec[b, : t.shape[0]] = t
                clen = [t.shape[0] for t in toks]
                lp = m.engine.lm_logprobs(ref_outs, lens, codec, clen).cpu()
                for b in range(B):
                    seq = orc.llm_input(ref_outs[b, : lens[b]], toks[b])
                    ref = orc.lm_score_all(seq, 1 + lens[b])
                    print(f" lm_logprobs[{b}]: {err(lp[b, : ref.shape[0]], ref)}", flush=True)
                # probe the LM stack on utterance 0 alone
                seq0 = orc.llm_input(ref_outs[0, : lens[0]], toks[0])
                mask0 = orc.lm_mask(seq0.size(0), 1 + lens[0])
                probe_stack(m, orc.codec_lm, 1, lambda: m.engine.lm_logprobs(ref_outs[:1, : lens[0]], lens[:1], codec[:1, : clen[0]], clen[:1]),
                            seq0.unsqueeze(0), mask0, [seq0.size(0)], None)
            except Exception:
                traceback.print_exc()
            # ---- decode, forced (step form) and free greedy
            try:
                forced = torch.zeros(B, 6, spec.predict_nq, dtype=torch.int64)
                for b, t in enumerate(toks):
                    forced[b, : t.shape[0]] = t
                tk, ol, slp = m.engine.decode_codec(ref_outs, lens, 6, sampling=False, forced=forced, return_logp=True)
                for b in range(B):
                    _, ref = orc.decode_codec(ref_outs[b, : lens[b]], 6, return_logp=True)
                    for s in range(ref.shape[0]):
                        print(f" step[{b}][{s}]: {err(slp[b, s], ref[s])}", flush=True)
                tk2, ol2 = m.engine.decode_codec(ref_outs, lens, 6, sampling=False)
                for b in range(B):
                    print(f" greedy[{b}]: engine {tk2[b, :ol2[b]].cpu().tolist()}  oracle {toks[b].tolist()}", flush=True)
            except Exception:
                traceback.print_exc()
            # ---- codec embedding
            try:
                embs = m.engine.codec_emb(ref_outs, lens, codec, clen).cpu()
                ref = orc.cal_codec_emb([ref_outs[b, : lens[b]] for b in range(B)], toks)
                for b in range(B):
                    print(f" codec_emb[{b}]: {err(embs[b, : clen[b]], ref[b])}", flush=True)
            except Exception:
                traceback.print_exc()


if __name__ == "__main__":
    main()
