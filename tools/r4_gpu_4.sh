#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export FC_WAIVER_JSON=$OUT/tie_waivers_freq.json
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_laura.py -m gpu -q -k "freq_codec or top_k_with or wrong_shapes" > $OUT/freq_pytest.log 2>&1; echo "rc=$?" >> $OUT/freq_pytest.log
tail -40 $OUT/freq_pytest.log
