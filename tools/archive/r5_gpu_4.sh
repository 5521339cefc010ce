#!/bin/bash
# round 5, call 4: what the staging waves of a PLAIN layer wait for: timelines under FC_ABLATE (16 no weight DMA, 1 no MFMA, 4 no slab loads)
mkdir -p gpurun_out/r5
export FC_LIB=$PWD/funcodec_amd/libfuncodec_amd_timeline.so
for abl in 0 16 1 17 4 20; do
    echo "=== FC_ABLATE=$abl decoder.model.9.convtr 10000 (quad)"
    FC_ABLATE=$abl timeout 300 python tools/ablate_layer.py decoder.model.9.convtr 10000 2>&1 | grep -v "amdgpu.ids" | awk 'NR<=3 || /item  [5-9] |item 1[0-4] /'
done > gpurun_out/r5/timeline_abl.txt 2>&1
