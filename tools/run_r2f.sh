#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r2f_test.log
tail -4 gpurun_out/r2f_test.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2f_bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
for k in d['kernels']: print('%-62s n=%2d %7.3f ms %8.2f us  %6s TF %7s GB/s' % (k['kernel'][:62], k['launches_per_step'], k['ms_per_step'], k['avg_us_per_launch'], k['tflops'], k['alg_gbs']))
PY
FC_BM256=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-event-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FC_BM256=0 ms_per_step', d['ms_per_step'])"
