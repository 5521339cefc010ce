#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r2c_test.log
tail -4 gpurun_out/r2c_test.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c_bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
for k in d['kernels']: print('%-62s n=%2d %7.3f ms %8.2f us  %6s TF %7s GB/s' % (k['kernel'][:62], k['launches_per_step'], k['ms_per_step'], k['avg_us_per_launch'], k['tflops'], k['alg_gbs']))
PY
for a in 0 1 2 4 8 3 7; do FC_ABLATE_RH=$a python tools/ablate_reshead.py encoder.model.1 160000 2>&1 | grep reshead; done
for w in 256 512 1024 1536; do FC_RH_WGS=$w python tools/ablate_reshead.py encoder.model.1 160000 2>&1 | grep reshead; done
for a in 0 1 4; do FC_ABLATE_RH=$a python tools/ablate_reshead.py encoder.model.4 80000 2>&1 | grep reshead; done
