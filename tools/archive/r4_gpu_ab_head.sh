#!/bin/bash
# A/B of library builds on the HEADLINE step in one call: usage r4_gpu_ab_head.sh default <variant> ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" != "default" ]; then export FC_LIB=$R/funcodec_amd/libfc_$v.so; else unset FC_LIB; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_head_$v.json 2> $OUT/bench_head_$v.err
  python -c "
import json
d = json.load(open('gpurun_out/bench_head_$v.json')); print('$v', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us_per_launch'])" || tail -5 $OUT/bench_head_$v.err
done
done
unset FC_LIB
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lstm or golden" -x > $OUT/lstm_pytest.log 2>&1; tail -3 $OUT/lstm_pytest.log
