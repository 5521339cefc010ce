"""Profiling aid (not a test): time the bottleneck LSTM at benchmark shape under FC_ABLATE_LSTM variants.
usage: FC_ABLATE_LSTM=<mask> python tools/ablate_lstm.py [encoder|decoder] [T] [B]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
from helpers import engine_for, freq_engine_for

which = {"encoder": "encoder.model.16.lstm", "decoder": "decoder.model.1.lstm"}.get(sys.argv[1] if len(sys.argv) > 1 else "encoder", sys.argv[1] if len(sys.argv) > 1 else "")
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cfg_name = os.environ.get("FC_CFG", "ds640")
m = (freq_engine_for if cfg_name.startswith(("freq", "tinyfreq")) else engine_for)(cfg_name, 0)
eng = m.engine
if which in ("encoder", "decoder") or which not in {k.rsplit(".", 1)[0] for k in eng.expected_tensors()}:      # FreqCodec nets: other layer indices
    side = "decoder" if "decoder" in which else "encoder"
    which = sorted({k[:k.index(".lstm") + 5] for k in eng.expected_tensors() if ".lstm." in k and k.startswith(side)})[0]
H = eng.expected_tensors()[which + ".weight_hh_l0"][1]
x = torch.randn(B, H, T, device="cuda")
for _ in range(2):
    eng.lstm_forward(which, x)
torch.cuda.synchronize()
n = 10
eng.set_profiling(True)
for _ in range(n):
    eng.lstm_forward(which, x)
prof = [p for p in eng.read_profile() if p["launches"]]
for p in prof:
    if "lstm" in p["kernel"]:
        print(f"ablate={os.environ.get('FC_ABLATE_LSTM','0'):>3s} {which} T={T} B={B}: {p['kernel']} {p['total_ms'] * 1e3 / n:8.1f} us  ({p['total_ms'] * 1e3 / n / (T + 1):.2f} us/step)")
