#!/bin/bash
# round 5: the whole GPU suite + smoke + the default bench line (with its secondary measurements) in one call
mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r5/pytest_full.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r5/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r5/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], d["value"], "whole_step", d["whole_step"]["frac_of_f32_peak"])
print("roofline", {k:d["roofline"].get(k) for k in ("kernel","frac","traffic_over_algorithmic","algorithmic_bytes_per_launch","traffic")})
for k,v in d.get("secondary",{}).items(): print(k, {kk:v.get(kk) for kk in ("value","ms_per_step","decode_step_us","error")})
print("cpu", d.get("cpu_baseline",{}).get("value"))
P
