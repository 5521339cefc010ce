#!/bin/bash
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_conv_layer or padding_edge or random_shape_sweep or row_staging or staging_scheme or fused_resblock or e2e_against_reference_golden" > gpurun_out/r5/pytest_conv6.log 2>&1
tail -4 gpurun_out/r5/pytest_conv6.log
cp funcodec_amd/libfc_prev.so /tmp/ 2>/dev/null
bash tools/r5_gpu_ab.sh prev default
export FC_LIB=$PWD/funcodec_amd/libfuncodec_amd_timeline.so
for spec in "decoder.model.9.convtr 10000 elu" "encoder.model.15.conv 2000 elu" "decoder.model.3.convtr 250 elu"; do echo "=== wide $spec"; timeout 300 python tools/ablate_layer.py $spec 2>&1 | grep -v amdgpu.ids | awk "NR<=3 || /item  [5-9] |item 1[0-4] /"; done
