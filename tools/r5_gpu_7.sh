#!/bin/bash
mkdir -p gpurun_out/r5
for spec in "sc ds640 encoder.model.13.shortcut.conv 200 2" "e15 ds640 encoder.model.15.conv 200 2" "ct9 ds640 decoder.model.9.convtr 400 2" "e12 ds640 encoder.model.12.conv 400 2"; do
  set -- $spec
  FC_XQ=1 python tools/r5_layer_dump.py xq_$1 $2 $3 $4 $5 2>&1 | grep -v amdgpu
  FC_XQ=0 python tools/r5_layer_dump.py rq_$1 $2 $3 $4 $5 2>&1 | grep -v amdgpu
done
python - <<'P'
import numpy as np
for t in ("sc","e15","ct9","e12"):
    a=np.load(f"gpurun_out/r5/xq_{t}.npy"); b=np.load(f"gpurun_out/r5/rq_{t}.npy")
    d=np.abs(a-b)
    print(t, a.shape, "max", d.max())
    bad=d>1e-4
    print("  bad frac per utterance", bad.reshape(a.shape[0],-1).mean(1))
    print("  bad rows(channels) count", int(bad.any(axis=(0,2)).sum()), "of", a.shape[1], " first bad rows", np.nonzero(bad.any(axis=(0,2)))[0][:10])
    cols=np.nonzero(bad.any(axis=(0,1)))[0]
    print("  bad cols count", len(cols), "of", a.shape[2], " first", cols[:12], " last", cols[-6:])
P
