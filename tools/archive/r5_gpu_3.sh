#!/bin/bash
# round 5, call 3: per-phase timelines of one workgroup (FC_TIMELINE build), quad vs round-4 layout
mkdir -p gpurun_out/r5
export FC_LIB=$PWD/funcodec_amd/libfuncodec_amd_timeline.so
for q in 1 0; do
  for spec in "decoder.model.3.convtr 250" "decoder.model.9.convtr 10000" "decoder.model.4.block.1.conv 2000 elu" "encoder.model.15.conv 2000"; do
    echo "=== FC_QUAD=$q $spec"
    FC_QUAD=$q timeout 300 python tools/ablate_layer.py $spec 2>&1 | grep -v "amdgpu.ids" | head -60
  done
done > gpurun_out/r5/timeline.txt 2>&1
wc -l gpurun_out/r5/timeline.txt
