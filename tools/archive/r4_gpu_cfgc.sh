#!/bin/bash
# Config C shard shape (128 utterances x 10 s per rank, micro-batches of 32) under run-time switches, one call:
#   r4_gpu_cfgc.sh base:X=1 lds80:FC_SMALLN_LDS=80 ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $(echo $envs | tr ',' ' ') timeout 300 python - <<PY
import sys, os, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import torch
from helpers import engine_for, audio
m = engine_for("ds640", 0)
mb = int(os.environ.get("MB", "32"))
m.engine.micro_batch = mb
wav = audio(128, 160000, 1234, "noise").cuda()
for _ in range(2):
    m.engine.encode_decode(wav, 32, want_sub_quants=False)
torch.cuda.synchronize()
t = time.time()
n = 3
for _ in range(n):
    m.engine.encode_decode(wav, 32, want_sub_quants=False)
torch.cuda.synchronize()
ms = (time.time() - t) / n * 1e3
print("$name", round(ms, 2), "ms per 128 utterances =", round(1280 / ms * 1e3, 1), "audio-s/s")
PY
done
done
