#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for v in "" abl1 abl2; do
  if [ -n "$v" ]; then export FC_LIB=$R/funcodec_amd/libfc_$v.so; else unset FC_LIB; fi
  echo "== build: ${v:-default}"
  FC_LAURA_TRACE=$OUT/laura_trace_$v.bin timeout 300 python bench.py --workload laura --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_laura_$v.json 2> $OUT/bench_laura_$v.err
  python tools/laura_trace.py $OUT/laura_trace_$v.bin 2.1 | grep -v "^ATT\|^OUT"
  python -c "
import json
d = json.load(open('gpurun_out/bench_laura_$v.json')); print(d['decode_step_us'])" || tail -5 $OUT/bench_laura_$v.err
done
