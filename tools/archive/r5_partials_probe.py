"""Bring-up aid: read the GroupNorm partial sums a conv launch left in the workspace (fc_layer_forward allocation order: materialised input,
raw output, partials, affine) and compare them with the true per-(M tile, N tile) sums of the raw output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch, torch.nn.functional as F
from helpers import engine_for, oracle_for
prefix, T, B, BM, BN, S, k, padL, padR = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9])
m, orc = engine_for("ds640", 0), oracle_for("ds640", 0)
W = orc.sd[prefix + ".conv.weight"]; bias = orc.sd[prefix + ".conv.bias"]
cout, cin = W.shape[0], W.shape[1]
x = torch.randn(B, cin, T, generator=torch.Generator().manual_seed(5))
y = m.engine.layer_forward(prefix, x, apply_elu=True).cpu()
xin = F.pad(F.elu(x), (padL, padR), mode="reflect") if padL + padR else F.elu(x)
raw = F.conv1d(xin, W, bias, stride=S).double()
Tout = raw.shape[2]
ws = m.engine._ws
al = lambda o: (o + 255) & ~255
xq = os.environ.get("FC_XQ", "1") != "0"
Tp = padL + T + padR
off = al(256)
off += ((B * (cin // 4) * Tp + BN * S + k + 320) * 4 if xq else B * cin * T) * 4
off = al(off); off += B * cout * Tout * 4
off = al(off)
mt_n, nt_n = -(-cout // BM), -(-Tout // BN)
part = ws[off: off + B * mt_n * nt_n * 16].cpu().numpy().view(np.float64).reshape(B, mt_n, nt_n, 2)
for b in range(B):
    for mt in range(mt_n):
        for nt in range(nt_n):
            r = raw[b, mt * BM:(mt + 1) * BM, nt * BN:(nt + 1) * BN]
            print(f"b{b} mt{mt} nt{nt}: got sum {part[b, mt, nt, 0]:12.4f} sumsq {part[b, mt, nt, 1]:12.4f}   true {float(r.sum()):12.4f} {float((r * r).sum()):12.4f}")
