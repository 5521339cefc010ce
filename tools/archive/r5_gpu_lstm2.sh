#!/bin/bash
# round 5: the two-chain persistent LSTM (lstm_persist2_kernel) -- its parity tests, then an A / B against the single-barrier form on the
# headline step in the same call (one FC_AB_KNOBS build, FC_LSTM_CHAINS=1 selects the old form)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lstm or status or micro_batching or 32_utterance or hip_graph or full_size or e2e_against_reference_golden" > gpurun_out/r5/pytest_lstm2.log 2>&1
tail -5 gpurun_out/r5/pytest_lstm2.log
export FC_LIB=$R/funcodec_amd/libfc_ab.so
for rep in 1 2; do
for v in 1 2; do
  FC_LSTM_CHAINS=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r5/lstm_chains_$v.json 2> gpurun_out/r5/lstm_chains_$v.err
  python - <<P
import json
try:
    d = json.loads(open('gpurun_out/r5/lstm_chains_$v.json').read().strip().splitlines()[-1])
    ks = [k for k in d.get('kernels', []) if 'lstm' in k['kernel']]
    print('chains=$v rep $rep', d['ms_per_step'], [(k['launches_per_step'], k['ms_per_step'], k['avg_us_per_launch'], k['f32_frac']) for k in ks])
except Exception as e:
    print('chains=$v failed', e); print(open('gpurun_out/r5/lstm_chains_$v.err').read()[-800:])
P
done
done
