#!/bin/bash
# robustness sweeps after the FreqCodec kernel changes of round 4 (one gpurun call): random 2-D architectures, odd lengths, determinism
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out
FREQ=1 timeout 900 python tools/fuzz_archs.py 2000 2060 > $OUT/fuzz_freq.txt 2>&1; tail -1 $OUT/fuzz_freq.txt; grep CHECK $OUT/fuzz_freq.txt | head -5
timeout 900 python tools/fuzz_lengths.py > $OUT/fuzz_lengths.txt 2>&1; tail -2 $OUT/fuzz_lengths.txt
timeout 600 python - <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import torch
from helpers import freq_engine_for, audio
for cfg, seed, B, T in (("freqmpgr1", 0, 4, 16000), ("tinyfreqgr1", 7, 3, 2500), ("freqmp", 0, 2, 16000)):
    m = freq_engine_for(cfg, seed)
    wav, wav2 = audio(B, T, 99, "tones").cuda(), audio(B, T, 5, "noise").cuda()
    nq = m.arch.num_quantizers
    ref = {k: v.clone() for k, v in m.engine.encode_decode(wav, nq).items() if torch.is_tensor(v)}
    bad = 0
    for i in range(60):
        if i % 3 == 1:
            m.engine.encode_decode(wav2, nq)
        out = m.engine.encode_decode(wav, nq)
        bad += any(not torch.equal(out[k], v) for k, v in ref.items())
    print(f"determinism {cfg}: {bad} of 60 runs differ")
PY
