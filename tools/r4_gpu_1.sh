#!/bin/bash
# round-4 GPU call 1: LauraTTS persistent step (tests + A/B timing), FreqCodec real-audio fixtures with the reference variants
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export FC_WAIVER_JSON=$OUT/tie_waivers_laura.json
timeout 600 python -m pytest tests/test_laura.py -m gpu -q -x > $OUT/laura_pytest.log 2>&1; echo "laura rc=$?" >> $OUT/laura_pytest.log
tail -5 $OUT/laura_pytest.log
FC_LAURA_PERSIST=1 timeout 300 python bench.py --workload laura --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_laura_persist.json 2> $OUT/bench_laura_persist.err
FC_LAURA_PERSIST=0 timeout 300 python bench.py --workload laura --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_laura_chain.json 2> $OUT/bench_laura_chain.err
python - <<'PY'
import json
for n in ("persist", "chain"):
    try:
        d = json.load(open(f"gpurun_out/bench_laura_{n}.json"))
        print(n, d["ms_per_step"], d["decode_step_us"], d["phases_ms"], d.get("batch16"))
    except Exception as ex:
        print(n, "failed", ex, open(f"gpurun_out/bench_laura_{n}.err").read()[-800:])
PY
export FC_WAIVER_JSON=$OUT/tie_waivers_freq.json
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "freq_codec_against_reference_golden" > $OUT/freq_golden_pytest.log 2>&1; echo "freq rc=$?" >> $OUT/freq_golden_pytest.log
tail -30 $OUT/freq_golden_pytest.log
