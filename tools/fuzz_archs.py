"""One-off sweep (not a test): pseudo-random small architectures (config.py::fuzz_recipe_config) through the engine against the CPU
oracle.  usage: python tools/fuzz_archs.py FIRST LAST"""
import os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from helpers import audio, rms, index_report, state_for, freq_state_for
from torch_oracle import Oracle
from freq_oracle import FreqOracle
from funcodec_amd.model import EncodecMI355X

FREQ = os.environ.get("FREQ", "0") == "1"           # FREQ=1: fuzz_freq_recipe_config (2-D FreqCodec nets) instead of the 1-D codec

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    name = f"freqfuzz{seed}" if FREQ else f"fuzz{seed}"
    try:
        cfg, arch, sd = freq_state_for(name, seed) if FREQ else state_for(name, seed)
        tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
        m = EncodecMI355X(arch, "cuda:0")
        m.load_state_dict(tsd)
        orc = (FreqOracle if FREQ else Oracle)(cfg, tsd)
        B, T = 1 + seed % 3, 1200 + 211 * (seed % 17)
        ch = 2 if (not FREQ and arch.input_channels == 2) else 1
        wav = audio(B, T, 5000 + seed, "tones" if seed % 2 else "noise", ch)
        o = orc.inference(wav, None, True)
        r = m.inference(wav.cuda() if ch > 1 else wav.cuda().unsqueeze(1), bit_width=None, use_scale=True)
        m.engine.check_status()
        rep = index_report(r["code_indices"][0], o["code_indices"][0])
        ref = float(o["recon_speech"].double().pow(2).mean().sqrt()) if FREQ else 1.0
        w = rms(r["recon_speech"], o["recon_speech"]) / ref if not rep["frames_bad"] else float("nan")
        flag = "" if (rep["frames_bad"] <= max(1, rep["frames"] // 50) and not (w > (1e-3 if FREQ else 1e-4))) else "   <-- CHECK"
        bad += bool(flag)
        print(f"{name}: n_fft {arch.n_fft}/{arch.stft_hop} rf {arch.ratios_f} gr {arch.enc_conv_group_ratio} " if FREQ else "", end="")
        print(f"{name}: ratios {arch.ratios} nf {arch.n_filters} k {arch.kernel_size}/{arch.last_kernel_size}/{arch.residual_kernel_size} "
              f"res {arch.n_residual_layers}x{arch.dilation_base} lstm {arch.lstm_layers} {arch.norm}{' causal' if arch.causal else ''} "
              f"K {arch.codebook_size} nq {arch.num_quantizers}{' stereo' if arch.input_channels == 2 and not FREQ else ''}{' q0' if arch.q0_ds_ratio > 1 else ''}: frames_bad {rep['frames_bad']}/{rep['frames']} wav rms {w:.2e}{flag}", flush=True)
        del m
    except Exception as ex:
        bad += 1
        print(f"{name}: EXCEPTION {type(ex).__name__}: {str(ex)[:300]}   <-- CHECK", flush=True)
print("suspicious:", bad)
