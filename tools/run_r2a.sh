#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r2a_test.log
tail -4 gpurun_out/r2a_test.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2a_bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
for k in d['kernels']: print('%-62s n=%2d %7.3f ms %8.2f us  %6s TF %7s GB/s' % (k['kernel'][:62], k['launches_per_step'], k['ms_per_step'], k['avg_us_per_launch'], k['tflops'], k['alg_gbs']))
PY
python tools/ablate_lstm.py encoder.model.16.lstm 2>&1 | grep ablate
T=$R/funcodec_amd/libfuncodec_amd_timeline.so
FC_ABLATE=16 python tools/ablate_layer.py encoder.model.9.conv 40000 elu 2>&1 | grep ablate
FC_TARGET_WGS=256 python tools/ablate_layer.py encoder.model.9.conv 40000 elu 2>&1 | grep ablate
FC_LIB=$T python tools/ablate_layer.py encoder.model.9.conv 40000 elu > gpurun_out/tl_enc9.txt 2>&1
FC_LIB=$T python tools/ablate_layer.py encoder.model.7.shortcut.conv 40000 > gpurun_out/tl_enc7sc.txt 2>&1
FC_LIB=$T python tools/ablate_layer.py encoder.model.12.conv 10000 elu > gpurun_out/tl_enc12.txt 2>&1
FC_LIB=$T python tools/ablate_layer.py encoder.model.15.conv 2000 elu > gpurun_out/tl_enc15.txt 2>&1
head -60 gpurun_out/tl_enc9.txt
