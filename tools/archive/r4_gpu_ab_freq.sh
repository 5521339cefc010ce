#!/bin/bash
# A/B of library builds on the FreqCodec gr1 side measurement in one call: usage r4_gpu_ab_freq.sh default <variant> ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" != "default" ]; then export FC_LIB=$R/funcodec_amd/libfc_$v.so; else unset FC_LIB; fi
  timeout 300 python bench.py --workload freqcodec_gr1 --steps 5 --warmup 2 > $OUT/bench_freq_$v.json 2> $OUT/bench_freq_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_freq_$v.json"))
print("$v", d["value"], d["ms_per_step"], d["roofline_hbm"]["all_hbm_bound_conv_classes"]["frac"], [(k["ms_per_step"], k["kernel"][:24]) for k in d["kernels"] if "gconvtr" in k["kernel"]])
PY
done
done
unset FC_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "freq" 2>&1 | grep -v "^report\|^tie" | tail -3
