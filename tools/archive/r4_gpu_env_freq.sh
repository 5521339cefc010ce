#!/bin/bash
# A/B of run-time variants (environment switches) on the FreqCodec gr1 side measurement in one call:
#   r4_gpu_env_freq.sh base: fo4:FC_GCONV_FO3=4 ...     (NAME:ENV=VAL[,ENV=VAL]); the FreqCodec GPU tests then run under the LAST variant
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  timeout 300 env $(echo $envs | tr ',' ' ') python bench.py --workload freqcodec_gr1 --steps 5 --warmup 2 > $OUT/bench_freq_$name.json 2> $OUT/bench_freq_$name.err
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_freq_$name.json"))
ks = sorted(d["kernels"], key=lambda k: -k["ms_per_step"])
print("$name", d["value"], d["ms_per_step"], d["roofline_hbm"]["all_hbm_bound_conv_classes"]["frac"])
if $rep == 2: print("   ", [(round(k["ms_per_step"], 2), k["kernel"][:34]) for k in ks[:12]])
PY
done
done
last=${@: -1}; envs=${last#*:}
timeout 900 env $(echo $envs | tr ',' ' ') python -m pytest tests/test_gpu_parity.py -m gpu -q -k "freq" 2>&1 | grep -v "^report\|^tie" | tail -3
