#!/bin/bash
# NOTE: the kernels these switches select (FC_LSTM_ROLES / FC_LSTM_VARIANT / FC_LSTM_ASLEEP ...) are archived, not compiled: tools/experiments/lstm_two_roles.hip.txt
# (drop them back into csrc/kernels.hip with their launch wiring to re-run); kept as the record of how profiles/r06_lstm_two_role.txt was produced.
# round 6, call 4: two-role LSTM, sleep of chain A's gate wave between arrival and first poll (FC_LSTM_ASLEEP, units of 8 x 64 cycles; AB build)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r6/lstm
export FC_LIB=$R/funcodec_amd/libfc_ab.so
FC_LSTM_ROLES=0 timeout 200 python tools/ablate_lstm.py decoder 250 16 2>&1 | grep -v amdgpu.ids | cut -c1-40,118- | sed "s/^/single-role /"
for a in 0 2 4 6 8 12; do
  for b in 0 2; do
    FC_LSTM_ASLEEP=$a FC_LSTM_BSLEEP=$b timeout 200 python tools/ablate_lstm.py decoder 250 16 2>&1 | grep -v amdgpu.ids | cut -c1-40,118- | sed "s/^/asleep=$a bsleep=$b /"
  done
done
FC_CFG=freqmpgr1 FC_LSTM_ROLES=0 timeout 200 python tools/ablate_lstm.py encoder 500 32 2>&1 | grep -v amdgpu.ids | cut -c1-40,118- | sed "s/^/H512 single-role /"
for a in 2 4 6; do
  FC_CFG=freqmpgr1 FC_LSTM_ASLEEP=$a timeout 200 python tools/ablate_lstm.py encoder 500 32 2>&1 | grep -v amdgpu.ids | cut -c1-40,118- | sed "s/^/H512 asleep=$a /"
done
