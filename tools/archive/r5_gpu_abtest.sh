#!/bin/bash
# round 5: conv-layer + e2e parity subset, then an A/B of library builds on the headline step in the same call
#   usage: r5_gpu_abtest.sh <variant> ...    (variant "default" = funcodec_amd/libfuncodec_amd.so, else funcodec_amd/libfc_<variant>.so)
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_conv_layer or padding_edge or random_shape_sweep or row_staging or staging_scheme or fused_resblock or e2e_against_reference_golden or segmented" > gpurun_out/r5/pytest_conv.log 2>&1
tail -4 gpurun_out/r5/pytest_conv.log
bash tools/r5_gpu_ab.sh "$@"
