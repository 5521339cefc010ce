"""Probe (not a test): one encode_decode call of the benchmark batch captured into a HIP graph (torch.cuda.CUDAGraph captures the raw launches the
library enqueues on torch's stream) and replayed, against the same call launched kernel by kernel."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from helpers import engine_for, audio

m = engine_for("ds640", 0)
eng = m.engine
wav = audio(16, 160000, 1234).cuda()
for _ in range(3):
    ref = eng.encode_decode(wav, 32)
torch.cuda.synchronize()

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print(f"stream launches: {timeit(lambda: eng.encode_decode(wav, 32)):.3f} ms / step")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.encode_decode(wav, 32)            # warm-up on the side stream (workspace for this stream)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = eng.encode_decode(wav, 32)
g.replay()
torch.cuda.synchronize()
print("graph result equals stream result:", torch.equal(out["codes"], ref["codes"]), torch.equal(out["recon"], ref["recon"]))
print(f"graph replay   : {timeit(g.replay):.3f} ms / step")
eng.check_status()
