#!/bin/bash
mkdir -p gpurun_out/r5
for spec in "a ds640 encoder.model.13.shortcut.conv 100 2" "b ds640 encoder.model.13.shortcut.conv 100 1" "c ds640 encoder.model.12.conv 3000 2" "d ds640 encoder.model.12.conv 400 1" "e ds640 decoder.model.9.convtr 100 1" "f ds640 encoder.model.15.conv 200 1"  "g ds640 encoder.model.15.conv 1600 1"; do
  set -- $spec
  FC_XQ=1 python tools/r5_layer_dump.py xq_$1 $2 $3 $4 $5 2>&1 | grep -v amdgpu
  FC_XQ=0 python tools/r5_layer_dump.py rq_$1 $2 $3 $4 $5 2>&1 | grep -v amdgpu
done
python - <<'P'
import numpy as np
for t in "abcdefg":
    a=np.load(f"gpurun_out/r5/xq_{t}.npy"); b=np.load(f"gpurun_out/r5/rq_{t}.npy")
    d=np.abs(a-b); bad=d>1e-4
    cols=np.nonzero(bad.any(axis=(0,1)))[0]
    print(t, a.shape, "max", d.max(), "bad frac per utt", bad.reshape(a.shape[0],-1).mean(1), "bad cols", len(cols), cols[:8], cols[-4:])
P
