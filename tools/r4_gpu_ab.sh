#!/bin/bash
# A/B of library builds in ONE call (boxes differ by more than a few percent): usage r4_gpu_ab.sh <variant> ... ("" = the product library)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" != "default" ]; then export FC_LIB=$R/funcodec_amd/libfc_$v.so; else unset FC_LIB; fi
  timeout 300 python bench.py --workload laura --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_laura_$v.json 2> $OUT/bench_laura_$v.err
  python -c "
import json
d = json.load(open('gpurun_out/bench_laura_$v.json')); print('$v', d['decode_step_us'], d['batch16']['ms_per_step'])" || tail -5 $OUT/bench_laura_$v.err
done
done
