"""Not a test: a diagnostic sweep that keeps going after failures (used while bringing kernels up)."""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import torch.nn.functional as F

from helpers import *  # noqa


def section(name):
    print(f"\n=== {name} ===", flush=True)


def main():
    print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))
    man = manifest()
    # ---- per-layer conv checks on the tiny and ds640 plans
    for cfg_name, seed, B, T0 in (("tiny", 7, 3, 203), ("ds640", 0, 2, 3200)):
        section(f"layers {cfg_name}")
        m = engine_for(cfg_name, seed)
        orc = oracle_for(cfg_name, seed)
        et = m.engine.expected_tensors()
        prefixes = sorted({k.rsplit(".norm.weight", 1)[0] for k in et if k.endswith(".norm.weight")})
        for p in prefixes:
            try:
                tr = p.endswith("convtr")
                w = orc.sd[p + (".convtr.weight" if tr else ".conv.weight")]
                cin = w.shape[0] if tr else w.shape[1]
                k = w.shape[2]
                # pick a T appropriate to the layer's rate
                T = max(3, T0 // max(1, cin // 8)) if cfg_name == "tiny" else max(7, 4 * T0 // cin)
                x = torch.randn(B, cin, T, generator=torch.Generator().manual_seed(5))
                for elu in (False, True):
                    xin = F.elu(x) if elu else x
                    if tr:
                        stride = k // 2
                        ref = __import__("torch_oracle").sconvtr1d(xin, *orc._p(p), stride, orc.eps)
                    else:
                        # stride from the plan: down convs have k = 2*stride (k even), others stride 1
                        stride = k // 2 if (k % 2 == 0 and k > 1) else 1
                        ref = __import__("torch_oracle").sconv1d(xin, *orc._p(p), stride, orc.eps)
                    got = m.engine.layer_forward(p, x, apply_elu=elu).cpu()
                    err = (got - ref).abs().max().item() if got.shape == ref.shape else float("nan")
                    print(f"{p:40s} elu={int(elu)} x{tuple(x.shape)} -> {tuple(got.shape)} ref{tuple(ref.shape)} maxerr={err:.3e}", flush=True)
            except Exception:
                traceback.print_exc()
        if m.arch.lstm_layers:
            for p in [k.rsplit(".weight_ih_l0", 1)[0] for k in et if k.endswith(".weight_ih_l0")]:
                try:
                    H = et[p + ".weight_ih_l0"][1]
                    x = torch.randn(B, H, 9, generator=torch.Generator().manual_seed(6))
                    ref = orc._slstm(x, p)
                    got = m.engine.lstm_forward(p, x).cpu()
                    print(f"{p:40s} lstm x{tuple(x.shape)} maxerr={(got-ref).abs().max().item():.3e}", flush=True)
                except Exception:
                    traceback.print_exc()
    # ---- RVQ goldens
    section("rvq goldens")
    from funcodec_amd.model import EncodecMI355X
    for name in ("rvq_flat", "rvq_decay08"):
        try:
            c = man["cases"][name]
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
            print(name, rep, "quant rms", rms(quant.reshape(8, 250, 128), g["quantized"]), flush=True)
        except Exception:
            traceback.print_exc()
    # ---- end-to-end goldens
    section("e2e goldens")
    for name, c in man["cases"].items():
        if c.get("kind") == "rvq":
            continue
        try:
            m = engine_for(c["config"], c["weight_seed"], c["codebook_decay"])
            wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"])
            g = golden(name)
            n_q = c["n_q"]
            r = m.engine.encode(wav, n_q, want_enc_out=True)
            print(name, "enc_out rms", rms(r["enc_out"], g["encoder_out"]), "scale err",
                  float((r["scale"].cpu() - torch.from_numpy(g["scale"])).abs().max()), flush=True)
            rep = index_report(r["codes"], g["indices"].astype(np.int64))
            print("   indices", {k: v for k, v in rep.items() if k != "first_stage"}, rep["first_stage"][:10])
            r2 = m.engine.encode_decode(wav, n_q, use_scale=True)
            print("   recon rms", rms(r2["recon"], g["recon"]), "ref rms level", float(torch.from_numpy(g["recon"]).pow(2).mean().sqrt()))
            tok = torch.from_numpy(g["indices"].astype(np.int64)).permute(1, 2, 0).contiguous()
            w2, _ = m.engine.decode_codes(tok)
            print("   decode(ref codes) rms", rms(w2, g["recon_from_codes"]), flush=True)
        except Exception:
            traceback.print_exc()
    # ---- timing at config B
    section("timing ds640 B=16 T=160000")
    try:
        m = engine_for("ds640", 0)
        wav = audio(16, 160000, 1234).cuda()
        for it in range(3):
            torch.cuda.synchronize()
            t = time.time()
            r = m.engine.encode_decode(wav, 32)
            torch.cuda.synchronize()
            print("iter", it, "ms", (time.time() - t) * 1e3, flush=True)
        print("finite:", bool(torch.isfinite(r["recon"]).all()))
    except Exception:
        traceback.print_exc()


if __name__ == "__main__":
    main()
