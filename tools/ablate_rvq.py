"""Profiling aid (not a test): time the fused residual quantiser at benchmark shape under FC_ABLATE_RVQ variants."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import engine_for

m = engine_for("ds640", 0)
x = torch.randn(4000, 128, device="cuda")
for _ in range(3):
    m.engine.rvq_encode(x, 32)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    m.engine.rvq_encode(x, 32)
e1.record()
torch.cuda.synchronize()
print(f"ablate={os.environ.get('FC_ABLATE_RVQ', '0'):>3s} two={os.environ.get('FC_RVQ_TWO', '0')}: {e0.elapsed_time(e1) * 100:.1f} us per call (incl. host wrapper)")
