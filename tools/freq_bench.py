"""FreqCodec (BASELINE.json configs[3] shape: batch 64, recipe net) timing + per-kernel-class profile through the engine."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from funcodec_amd.config import freq_recipe_config; from funcodec_amd.synth import make_freq_state_dict
from funcodec_amd.config import arch_from_config
from funcodec_amd.model import EncodecMI355X
from funcodec_amd.synth import synthetic_audio

B, SEC, MB = int(os.environ.get("B", 64)), float(os.environ.get("SEC", 10)), int(os.environ.get("MB", 16))
cfg = freq_recipe_config(os.environ.get("CFG", "freqmp"))
arch = arch_from_config(cfg)
m = EncodecMI355X(arch, "cuda:0")
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_freq_state_dict(cfg, 0).items()})
m.engine.micro_batch = MB
T = int(SEC * 16000)
wav = torch.from_numpy(synthetic_audio(B, T, 1, "tones")).cuda()
print("workspace GB", m.engine.lib.fc_engine_workspace_bytes(m.engine._h, min(B, MB), T) / 1e9, flush=True)
for _ in range(2):
    m.engine.encode_decode(wav, arch.num_quantizers)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 5
for _ in range(N):
    r = m.engine.encode_decode(wav, arch.num_quantizers)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"B={B} T={T} micro={MB}: {dt*1e3:.2f} ms/step  {B*SEC/dt:.0f} audio-s/s", flush=True)
m.engine.set_profiling(True)
m.engine.read_profile()
m.engine.encode_decode(wav, arch.num_quantizers)
prof = [p for p in m.engine.read_profile() if p["launches"]]
tot = sum(p["total_ms"] for p in prof)
for p in sorted(prof, key=lambda p: -p["total_ms"]):
    tf = p["flops"] / p["total_ms"] / 1e9 if p["total_ms"] else 0
    gb = p["bytes"] / p["total_ms"] / 1e6 if p["total_ms"] else 0
    print(f"  {p['kernel']:<28} {p['total_ms']:8.3f} ms {100*p['total_ms']/tot:5.1f}%  n={p['launches']:4d}  {tf:7.1f} TFLOP/s {gb:7.0f} GB/s")
print(f"  sum {tot:.2f} ms")
m.engine.check_status()
