#!/bin/bash
# round 5: single-barrier vs two-chain persistent LSTM at H = 512 (FreqCodec: two batch tiles side by side, latency-bound) and H = 1024
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export FC_LIB=$R/funcodec_amd/libfc_ab.so
for rep in 1 2; do for c in 1 2; do
  FC_CFG=freqmpgr1 FC_LSTM_CHAINS=$c timeout 120 python tools/ablate_lstm.py encoder 251 32 2>&1 | grep -v amdgpu.ids | sed "s/^/H512 B32 chains=$c /"
  FC_CFG=freqmpgr1 FC_LSTM_CHAINS=$c timeout 120 python tools/ablate_lstm.py encoder 251 16 2>&1 | grep -v amdgpu.ids | sed "s/^/H512 B16 chains=$c /"
done; done
