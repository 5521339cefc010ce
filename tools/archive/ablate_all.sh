#!/bin/bash
# One-call breakdown of the hot kernels under their ablation masks (tuning aid; run through gpurun from the repo root).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ablate.txt
: > $OUT
for a in 0 1 2 8 4 11; do FC_ABLATE_RVQ=$a python $R/tools/ablate_rvq.py >> $OUT 2>&1; done
for a in 0 1 2 16 32 3; do FC_ABLATE_LSTM=$a python $R/tools/ablate_lstm.py encoder.model.16.lstm >> $OUT 2>&1; done
for a in 0 1 4 16 128; do FC_ABLATE=$a python $R/tools/ablate_layer.py encoder.model.15.conv 2000 elu >> $OUT 2>&1; done
for a in 0 1 4 16; do FC_ABLATE=$a python $R/tools/ablate_layer.py encoder.model.12.conv 10000 elu >> $OUT 2>&1; done
for a in 0 1 4 2; do FC_ABLATE=$a python $R/tools/ablate_layer.py encoder.model.7.shortcut.conv 40000 >> $OUT 2>&1; done
for a in 0 1 4; do FC_ABLATE=$a python $R/tools/ablate_layer.py encoder.model.9.conv 40000 elu >> $OUT 2>&1; done
for a in 0 1 4 2; do FC_ABLATE=$a python $R/tools/ablate_layer.py encoder.model.1.block.1.conv 160000 elu >> $OUT 2>&1; done
grep -v "^$" $OUT | grep -v amdgpu.ids
