"""One-off check (not a test): a long utterance (default 120 s, B = 1) through the engine against the CPU oracle --
exercises many-tile rows (7 500 tiles on the thin layers), a 3 000-step LSTM recurrence and 32-bit offset arithmetic."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import time
import torch
from helpers import audio, engine_for, index_report, oracle_for, rms

secs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
cfg = sys.argv[2] if len(sys.argv) > 2 else "ds640"
m = engine_for(cfg, 0)
wav = audio(1, 16000 * secs, 7, "tones")
t0 = time.time()
r = m.engine.encode_decode(wav.cuda(), 32)
torch.cuda.synchronize()
t1 = time.time()
o = oracle_for(cfg, 0).inference(wav, bit_width=None, use_scale=True)
t2 = time.time()
rep = index_report(r["codes"], o["code_indices"][0].numpy())
print(f"{cfg} {secs} s: engine {t1 - t0:.2f} s (incl. workspace alloc), oracle {t2 - t1:.1f} s; indices {rep}; "
      f"recon rms err {rms(r['recon'], o['recon_speech']):.3e}")
