"""Profiling aid (not a test): the per-class kernel table of the whole headline step (ds640, 16 x 10 s, n_q = 32) under one FC_ABLATE mask
(needs a library built with FC_BUILD_DEFINES=FC_AB_KNOBS, loaded through FC_LIB; results under a mask are garbage by design).
usage: FC_LIB=... FC_ABLATE=<mask> python tools/ablate_step.py <out.json>"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import engine_for, audio

m = engine_for("ds640", 0)
eng = m.engine
wav = audio(16, 160000, 1234, "tones").cuda()
for _ in range(2):
    eng.encode_decode(wav, 32)
torch.cuda.synchronize()
n = 5
eng.set_profiling(True)
for _ in range(n):
    eng.encode_decode(wav, 32)
rows = {}
for p in eng.read_profile():
    if p["launches"]:
        rows[p["kernel"]] = dict(launches_per_step=p["launches"] // n, us_per_launch=p["total_ms"] * 1e3 / p["launches"], ms_per_step=p["total_ms"] / n,
                                 tflops=(p["flops"] / p["total_ms"] / 1e9) if p["total_ms"] else 0.0, alg_tbs=(p["bytes"] / p["total_ms"] / 1e9) if p["total_ms"] else 0.0)
json.dump(dict(mask=int(os.environ.get("FC_ABLATE", "0")), classes=rows), open(sys.argv[1], "w"))
print("mask", os.environ.get("FC_ABLATE", "0"), "conv ms/step", round(sum(v["ms_per_step"] for k, v in rows.items() if k.startswith(("conv", "reshead"))), 3))
