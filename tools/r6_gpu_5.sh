#!/bin/bash
# NOTE: the kernels these switches select (FC_LSTM_ROLES / FC_LSTM_VARIANT / FC_LSTM_ASLEEP ...) are archived, not compiled: tools/experiments/lstm_two_roles.hip.txt
# (drop them back into csrc/kernels.hip with their launch wiring to re-run); kept as the record of how profiles/r06_lstm_two_role.txt was produced.
# round 6, call 5: two-role LSTM variants (AB build): P in chain B (variant 0) vs P in chain A through an LDS ring (variant 1), sleeps and poll gaps
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export FC_LIB=$R/funcodec_amd/libfc_ab.so
run() { timeout 200 python tools/ablate_lstm.py decoder 250 16 2>&1 | grep -v amdgpu.ids | grep -o "([0-9.]* us/step)" | tr '\n' ' '; }
echo "single-role: $(FC_LSTM_ROLES=0 run)"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lstm" 2>&1 | tail -2
FC_LSTM_VARIANT=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lstm" 2>&1 | tail -2
for v in 1 0; do
  for a in 0 2 4 6; do
    for b in 0 2 4 6; do
      for g in 1 4; do
        echo "variant=$v asleep=$a bsleep=$b gap=$g: $(FC_LSTM_VARIANT=$v FC_LSTM_ASLEEP=$a FC_LSTM_BSLEEP=$b FC_LSTM_POLLGAP=$g run)"
      done
    done
  done
done
