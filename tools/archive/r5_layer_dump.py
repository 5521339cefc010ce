"""Bring-up aid: one layer through the C ABI, output saved to gpurun_out/r5/<tag>.npy (compare two builds / env settings offline)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import engine_for
tag, cfg, prefix, T, B = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
m = engine_for(cfg, 0)
et = m.engine.expected_tensors()
tr = prefix.endswith("convtr")
w = et[prefix + (".convtr.weight" if tr else ".conv.weight")]
cin = w[0] if tr else w[1]
x = torch.randn(B, cin, T, generator=torch.Generator().manual_seed(5))
y = m.engine.layer_forward(prefix, x, apply_elu=True).cpu().numpy()
np.save(f"gpurun_out/r5/{tag}.npy", y)
print(tag, y.shape, float(np.abs(y).max()))
