#!/bin/bash
# full GPU suite + default bench (the round's checkpoint run)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export FC_WAIVER_JSON=$OUT/tie_waivers.json
timeout 1200 python -m pytest tests -m gpu -q > $OUT/all_pytest.log 2>&1; echo "rc=$?" >> $OUT/all_pytest.log
tail -25 $OUT/all_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["whole_step"])
print({k: (v.get("value"), v.get("ms_per_step"), v.get("decode_step_us")) if isinstance(v, dict) else v for k, v in d.get("secondary", {}).items()})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
python __graft_entry__.py smoke 2>&1 | tail -3
