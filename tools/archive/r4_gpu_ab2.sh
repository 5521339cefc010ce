#!/bin/bash
# LauraTTS A/B in one call over (library, env) variants: "name:lib:ENV=VAL,ENV=VAL"
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; envs=${rest#*:}
  (
  if [ "$lib" != "default" ]; then export FC_LIB=$R/funcodec_amd/libfc_$lib.so; fi
  IFS=',' read -ra kv <<< "$envs"
  for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
  timeout 300 python bench.py --workload laura --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_laura_$name.json 2> $OUT/bench_laura_$name.err
  python -c "
import json
d = json.load(open('gpurun_out/bench_laura_$name.json')); print('$name', d['decode_step_us'], d['batch16']['ms_per_step'])" || tail -5 $OUT/bench_laura_$name.err
  )
done
done
