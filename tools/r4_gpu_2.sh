#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export FC_WAIVER_JSON=$OUT/tie_waivers_laura.json
timeout 600 python -m pytest tests/test_laura.py -m gpu -q -x > $OUT/laura_pytest.log 2>&1; tail -4 $OUT/laura_pytest.log
FC_LAURA_TRACE=$OUT/laura_trace.bin timeout 300 python bench.py --workload laura --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_laura_persist.json 2> $OUT/bench_laura_persist.err
python tools/laura_trace.py $OUT/laura_trace.bin 2.1
python - <<'PY'
import json
for n in ("persist",):
    try:
        d = json.load(open(f"gpurun_out/bench_laura_{n}.json"))
        print(n, d["ms_per_step"], d["decode_step_us"], d["phases_ms"], d.get("batch16"))
    except Exception as ex:
        print(n, "failed", ex, open(f"gpurun_out/bench_laura_{n}.err").read()[-800:])
PY
