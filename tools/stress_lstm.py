"""Stress aid (not a test): alternate two inputs through the persistent LSTM with nothing else in between (the
hidden-state history lives at the same addresses in every launch) and compare each result with its first value."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import engine_for

cfg = sys.argv[1] if len(sys.argv) > 1 else "ds320"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = int(sys.argv[3]) if len(sys.argv) > 3 else 50
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
m = engine_for(cfg, 0)
eng = m.engine
p = [k[: -len(".weight_hh_l0")] for k in eng.expected_tensors() if k.endswith(".weight_hh_l0")][0]
H = eng.expected_tensors()[p + ".weight_hh_l0"][1]
xs = [torch.randn(B, H, T, device="cuda", generator=torch.Generator("cuda").manual_seed(s)) for s in (1, 2, 3)]
refs = [eng.lstm_forward(p, x).clone() for x in xs]
bad = 0
first = None
for i in range(iters):
    j = (i * 7 + i // 3) % 3
    y = eng.lstm_forward(p, xs[j])
    if not torch.equal(y, refs[j]):
        bad += 1
        if first is None:
            first = (i, float((y - refs[j]).abs().max()))
print(f"{cfg} {p} B={B} T={T} iters={iters}: mismatching runs {bad} first={first}")
