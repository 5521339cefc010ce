#!/bin/bash
# round 6: the whole GPU suite + smoke on the current build (what the driver runs at round end), then the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r6/pytest_full.log 2>&1
tail -12 gpurun_out/r6/pytest_full.log | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r6/bench_default.json 2> gpurun_out/r6/bench_default.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r6/bench_default.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["kernel"][:40], d["roofline"]["frac"],
      "conv", d["roofline"].get("conv_class", {}).get("frac"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("port_over_reference", {}).get("ratio"))
for k, v in d.get("secondary", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), (v.get("whole_step") or {}).get("traffic_over_algorithmic"), (v.get("roofline_hbm") or {}).get("traffic"))
print("accounting_errors", d.get("accounting_errors"))
P
