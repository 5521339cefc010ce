#!/bin/bash
# round 6, call 7: workgroup-count selection by the tile-balance model (default) against the fixed 512-workgroup target (FC_TARGET_WGS=512 = rounds 1-5)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r6/gsel
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_conv_layer or staging_scheme or e2e_against_reference_golden or freq_codec_against" 2>&1 | tail -2
for v in new old new old; do
  if [ $v = old ]; then export FC_TARGET_WGS=512; else unset FC_TARGET_WGS; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r6/gsel/bench_$v.json 2> gpurun_out/r6/gsel/bench_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r6/gsel/bench_$v.json')); print('$v', 'ms_per_step', d['ms_per_step'], [(k['kernel'][17:42], k['avg_us_per_launch']) for k in d['kernels'] if '5, 3' in k['kernel'] or '5, 4' in k['kernel']])"
done
for v in new old; do
  if [ $v = old ]; then export FC_TARGET_WGS=512; else unset FC_TARGET_WGS; fi
  timeout 600 python bench.py --workload freqcodec_gr1 --steps 3 --warmup 1 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v freqcodec', d['ms_per_step'])"
  timeout 600 python bench.py --workload laura --steps 2 --warmup 1 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v laura', d.get('ms_per_step'), d.get('decode_step_us'))"
done
