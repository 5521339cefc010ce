"""Profiling aid (not a test): time one plan layer at benchmark shape under FC_ABLATE variants
(HIP-event timing of the raw C-ABI call, 20 launches).
usage: FC_ABLATE=<mask> python tools/ablate_layer.py <prefix> <T> [elu]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import engine_for

prefix, T = sys.argv[1], int(sys.argv[2])
elu = int(len(sys.argv) > 3)
m = engine_for("ds640", 0)
eng = m.engine
et = eng.expected_tensors()
tr = prefix.endswith("convtr")
w = et[prefix + (".convtr.weight" if tr else ".conv.weight")]
cin, cout = (w[0], w[1]) if tr else (w[1], w[0])
B = 16
x = torch.randn(B, cin, T, device="cuda")
Tout = eng.lib.fc_layer_out_len(eng._h, prefix.encode(), T)
y = torch.empty(B, cout, Tout, device="cuda")
ws = torch.empty(B * (cin * T + cout * (Tout + 64)) * 4 * 2 + (64 << 20), dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call():
    rc = eng.lib.fc_layer_forward(eng._h, prefix.encode(), C.c_void_p(x.data_ptr()), B, T, elu, C.c_void_p(y.data_ptr()),
                                  C.c_void_p(ws.data_ptr()), ws.numel(), st)
    assert rc == 0, eng.lib.fc_last_error()


for _ in range(3):
    call()
torch.cuda.synchronize()
n = 20
eng.set_profiling(True)
for _ in range(n):
    call()
prof = [p for p in eng.read_profile() if p["launches"]]
us = sum(p["total_ms"] for p in prof if p["kernel"].startswith("conv")) * 1e3 / n
print(f"ablate={os.environ.get('FC_ABLATE','0'):>3s} {prefix} T={T} elu={elu}: conv kernel {us:8.1f} us")

tl = (C.c_ulonglong * (2 * 24 * 8))()
if eng.lib.fc_debug_timeline(C.cast(tl, C.c_void_p)) == 0 and any(tl):
    import numpy as np
    a = np.array(list(tl), dtype=np.int64).reshape(2, 24, 8)
    t0 = a[a > 0].min()
    rt = a[0, :, 7]
    n = int((rt > 0).sum())
    if n > 1:
        print(f"calibration: {(a[0, n - 1, 0] - a[0, 0, 0]) / ((rt[n - 1] - rt[0]) * 10.0):.3f} s_memtime ticks per ns (s_memrealtime = 100 MHz)")
    print("matrix role  : item start | +mainloop | +epilogue | +barrier   (shader ticks, 100 MHz => x10 ns)")
    for f in range(24):
        r = a[0, f]
        if r[0]:
            print(f"  item {f:2d} @ {r[0] - t0:7d}: {r[1] - r[0]:6d} {r[2] - r[1]:6d} {r[3] - r[2]:6d}")
    print("staging role : item start | +write_slab | +load issue | +barrier | +flush")
    for f in range(24):
        r = a[1, f]
        if r[0]:
            print(f"  item {f:2d} @ {r[0] - t0:7d}: {r[1] - r[0]:6d} {r[2] - r[1]:6d} {r[3] - r[2]:6d} {r[4] - r[3]:6d}")
