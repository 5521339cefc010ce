#!/bin/bash
# round 6, call 10: LDS bank conflicts of the row staging's slab stores: 4 columns per lane (default) vs 2 / 1 (FC_ROW_CW, FC_AB_KNOBS build)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r6/cw
export FC_LIB=$R/funcodec_amd/libfc_ab.so
for cw in 1 2; do
  FC_ROW_CW=$cw timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_conv_layer or e2e_against_reference_golden[ds640 or padding_edge" 2>&1 | tail -1
done
for cw in 4 1 2 4 1 2; do
  FC_ROW_CW=$cw timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r6/cw/bench_$cw.json 2> gpurun_out/r6/cw/bench_$cw.err
  python -c "
import json; d=json.load(open('gpurun_out/r6/cw/bench_$cw.json')); print('cw=$cw ms_per_step', d['ms_per_step'], [(k['kernel'][17:46], k['avg_us_per_launch']) for k in d['kernels'] if ', true, true' in k['kernel'] and k['ms_per_step'] > 0.2])"
done
