#!/bin/bash
# round 5: calibration of the two-chain LSTM -- step time of both forms with and without the grid barriers, the MFMA chain microbenchmark
# (2 vs 4 accumulators), and compile-time ablation builds (funcodec_amd/libfc_abl<mask>.so, FC_LSTM_ABL) when present
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export FC_LIB=$R/funcodec_amd/libfc_ab.so
for c in 1 2; do for a in 0 1; do
  FC_LSTM_CHAINS=$c FC_ABLATE_LSTM=$a timeout 120 python tools/ablate_lstm.py decoder 250 16 2>&1 | grep -v amdgpu.ids | sed "s/^/chains=$c /"
done; done
for f in funcodec_amd/libfc_abl*.so; do
  [ -f "$f" ] || continue
  for c in 1 2; do for a in 0 1; do
    FC_LIB=$R/$f FC_LSTM_CHAINS=$c FC_ABLATE_LSTM=$a timeout 120 python tools/ablate_lstm.py decoder 250 16 2>&1 | grep -v amdgpu.ids | cut -c1-40,118- | sed "s|^|$(basename $f) chains=$c |"
  done; done
done
timeout 60 tests/micro/bin/mfma_16x16x4_chain 2>&1 | tail -8
