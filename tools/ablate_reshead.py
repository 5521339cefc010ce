"""Profiling aid (not a test): time the fused res-block head at benchmark shape under FC_ABLATE_RH variants.
usage: FC_ABLATE_RH=<mask> python tools/ablate_reshead.py [prefix] [T]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import engine_for

prefix = sys.argv[1] if len(sys.argv) > 1 else "encoder.model.1"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 160000
m = engine_for("ds640", 0)
eng = m.engine
C = eng.expected_tensors()[prefix + ".shortcut.conv.conv.bias"][0]
x = torch.randn(16, C, T, device="cuda")
for _ in range(3):
    eng.resblock_forward(prefix, x)
torch.cuda.synchronize()
n = 10
eng.set_profiling(True)
for _ in range(n):
    eng.resblock_forward(prefix, x)
for p in eng.read_profile():
    if p["launches"]:
        print(f"ablate_rh={os.environ.get('FC_ABLATE_RH', '0'):>2s} wgs={os.environ.get('FC_RH_WGS', '-'):>4s} {prefix} T={T}: {p['kernel']:44s} {p['total_ms'] * 1e3 / p['launches']:8.1f} us")
