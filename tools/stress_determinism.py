"""Stress aid (not a test): repeat the same calls many times and report any run whose output differs from the first
(a data race shows up as a non-deterministic result).  usage: python tools/stress_determinism.py [cfg] [B] [T] [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import engine_for, audio

cfg = sys.argv[1] if len(sys.argv) > 1 else "ds320"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 100
m = engine_for(cfg, 0)
wav = audio(B, T, 99, "tones").cuda()
wav2 = audio(B, T, 5, "noise").cuda()
ref = m.engine.encode_decode(wav, 32)
ref = {k: v.clone() for k, v in ref.items() if torch.is_tensor(v)}
bad = {}
for i in range(iters):
    if i % 3 == 1:
        m.engine.encode_decode(wav2, 32)          # different data through the same workspace in between
    out = m.engine.encode_decode(wav, 32)
    for k, v in ref.items():
        if not torch.equal(out[k], v):
            bad.setdefault(k, []).append(i)
print(f"{cfg} B={B} T={T} iters={iters} persist={os.environ.get('FC_LSTM_PERSIST', '1')}: mismatching runs per output:",
      {k: (len(v), v[:5]) for k, v in bad.items()} if bad else "none")
