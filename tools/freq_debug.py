"""FreqCodec path bring-up: engine vs oracle/freq_oracle.py + the reference's goldens, stage by stage."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from helpers import manifest, golden, audio, rms, index_report
from freq_oracle import FreqOracle
from funcodec_amd.config import freq_recipe_config; from funcodec_amd.synth import make_freq_state_dict
from funcodec_amd.config import arch_from_config
from funcodec_amd.model import EncodecMI355X

MAN = manifest()
for name in sys.argv[1:] or ["tinyfreq_b2_t2000", "freqmp_b1_t16000"]:
    c = MAN["cases"][name]
    cfg = freq_recipe_config(c["config"])
    sd = {k: torch.from_numpy(v) for k, v in make_freq_state_dict(cfg, c["weight_seed"]).items()}
    arch = arch_from_config(cfg)
    m = EncodecMI355X(arch, "cuda:0")
    m.load_state_dict(sd)
    wav = audio(c["batch"], c["samples"], c["audio_seed"], c["audio_kind"])
    g = golden(name)
    print(name, "frames", m.engine.frames(c["samples"]), "golden idx", g["indices"].shape, flush=True)
    r = m.engine.encode(wav, arch.num_quantizers, want_enc_out=True)
    m.engine.check_status()
    print(" enc_out rms", rms(r["enc_out"], g["encoder_out"]), "ref rms", float(np.sqrt((g["encoder_out"] ** 2).mean())))
    print(" index", index_report(r["codes"], g["indices"].astype(np.int64)))
    r2 = m.engine.encode_decode(wav, arch.num_quantizers, use_scale=True)
    m.engine.check_status()
    print(" recon", tuple(r2["recon"].shape), g["recon"].shape, "rms", rms(r2["recon"], g["recon"]), "ref rms", float(np.sqrt((g["recon"] ** 2).mean())))
    w3 = m.engine.decode_emb(torch.from_numpy(g["quantized"]))
    sc = torch.from_numpy(g["scale"]).view(-1, 1, 1)
    print(" decode(ref quantized)", tuple(w3.shape), rms(w3.cpu()[:, :, :c["samples"]] * sc, g["recon"]))
