import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from funcodec_amd.config import arch_from_config, recipe_config
from funcodec_amd.model import EncodecMI355X
from funcodec_amd.synth import make_freq_state_dict, synthetic_audio
cfg = recipe_config("freqmpgr1"); arch = arch_from_config(cfg)
m = EncodecMI355X(arch, "cuda:0"); m.load_state_dict({k: torch.from_numpy(v) for k, v in make_freq_state_dict(cfg, 0).items()})
eng = m.engine; eng.micro_batch = 32
B, T = int(sys.argv[1]), int(sys.argv[2])
wav = torch.from_numpy(synthetic_audio(B, T, 1234)).cuda()
r = eng.encode(wav, 32, want_enc_out=True)
print("enc_out finite", bool(torch.isfinite(r["enc_out"]).all()), "quantized finite", bool(torch.isfinite(r["quantized"]).all()))
r2 = eng.encode_decode(wav, 32, use_scale=True)
bad = ~torch.isfinite(r2["recon"])
print("recon finite", not bool(bad.any()), "bad utterances", bad.any(-1).any(-1).nonzero().flatten().tolist()[:8], "first bad sample", bad[0,0].nonzero().flatten()[:4].tolist() if bad[0].any() else None)
eng.check_status()
