#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export FC_WAIVER_JSON=$OUT/tie_waivers_freq.json
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "freq" > $OUT/freq_pytest.log 2>&1; echo "rc=$?" >> $OUT/freq_pytest.log
tail -8 $OUT/freq_pytest.log
for v in 1 0; do
  FC_GCONV_LDS3=$v timeout 300 python bench.py --workload freqcodec_gr1 --steps 5 --warmup 2 > $OUT/bench_freq_gr1_lds$v.json 2> $OUT/bench_freq_gr1_lds$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_freq_gr1_lds$v.json"))
print("lds3=$v", d["value"], d["ms_per_step"], d["roofline_hbm"]["all_hbm_bound_conv_classes"])
for k in d["kernels"][:12]:
    if "gconv" in k["kernel"]: print("   ", k["ms_per_step"], k["launches_per_step"], k["alg_gbs"], k["kernel"])
PY
done
