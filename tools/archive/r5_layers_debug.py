"""Round-5 bring-up aid: every conv layer and LSTM block of a recipe through the C ABI against the CPU oracle (max abs error per layer)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.nn.functional as F
from helpers import engine_for, oracle_for
import torch_oracle

cfg_name, T0 = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m, orc = engine_for(cfg_name, 0), oracle_for(cfg_name, 0)
et = m.engine.expected_tensors()
prefixes = [k.rsplit(".norm.weight", 1)[0] for k in et if k.endswith(".norm.weight")]
bad = 0
for p in prefixes:
    tr = p.endswith("convtr")
    w = orc.sd[p + (".convtr.weight" if tr else ".conv.weight")]
    cin, k = (w.shape[0] if tr else w.shape[1]), w.shape[2]
    T = max(7, 16 * T0 // max(cin, 16))
    x = torch.randn(B, cin, T, generator=torch.Generator().manual_seed(5))
    for elu in (False, True):
        xin = F.elu(x) if elu else x
        stride = k // 2 if (tr or (k % 2 == 0 and k > 1)) else 1
        ref = torch_oracle.sconvtr1d(xin, *orc._p(p), stride, orc.eps) if tr else torch_oracle.sconv1d(xin, *orc._p(p), stride, orc.eps)
        got = m.engine.layer_forward(p, x, apply_elu=elu).cpu()
        err = (got - ref).abs().max().item() if got.shape == ref.shape else float("nan")
        flag = "" if err < 5e-5 else "   <<<<<< BAD"
        bad += bool(flag)
        print(f"{p:38s} elu={int(elu)} x{tuple(x.shape)} -> {tuple(got.shape)} maxerr={err:.3e}{flag}", flush=True)
for p in [k.rsplit(".weight_ih_l0", 1)[0] for k in et if k.endswith(".weight_ih_l0")]:
    H = et[p + ".weight_ih_l0"][1]
    x = torch.randn(B, H, 37, generator=torch.Generator().manual_seed(6))
    ref = orc._slstm(x, p)
    got = m.engine.lstm_forward(p, x).cpu()
    err = (got - ref).abs().max().item()
    bad += err > 1e-5
    print(f"{p:38s} lstm x{tuple(x.shape)} maxerr={err:.3e}{'' if err < 1e-5 else '   <<<<<< BAD'}", flush=True)
m.engine.check_status()
print("BAD LAYERS:", bad)
