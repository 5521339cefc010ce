#!/bin/bash
# NOTE: the kernels these switches select (FC_LSTM_ROLES / FC_LSTM_VARIANT / FC_LSTM_ASLEEP ...) are archived, not compiled: tools/experiments/lstm_two_roles.hip.txt
# (drop them back into csrc/kernels.hip with their launch wiring to re-run); kept as the record of how profiles/r06_lstm_two_role.txt was produced.
# round 6, call 3: the two-role persistent LSTM (default) against the single-role kernel (FC_LSTM_ROLES=0) -- parity first, then the recurrence
# alone at the benchmark shape (B = 16, H = 1024, T = 250) and H = 512 (FreqCodec), then the headline bench of both in the same call
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r6/lstm
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lstm or e2e_against_reference_golden or micro_batching or long_utterance or timeout or barrier" > gpurun_out/r6/pytest_3.log 2>&1
tail -4 gpurun_out/r6/pytest_3.log
for r in 1 0 1 0; do
  FC_LSTM_ROLES=$r timeout 200 python tools/ablate_lstm.py decoder 250 16 2>&1 | grep -v amdgpu.ids | sed "s/^/roles=$r /"
done
for r in 1 0; do
  FC_LSTM_ROLES=$r timeout 200 python tools/ablate_lstm.py decoder 250 32 2>&1 | grep -v amdgpu.ids | sed "s/^/roles=$r B=32 /"
done
for r in 1 0 1 0; do
  FC_LSTM_ROLES=$r timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r6/lstm/bench_roles$r.json 2> gpurun_out/r6/lstm/bench_roles$r.err
  python -c "
import json; d=json.load(open('gpurun_out/r6/lstm/bench_roles$r.json')); print('roles=$r', 'ms_per_step', d['ms_per_step'], [(k['kernel'][:24], k['avg_us_per_launch']) for k in d['kernels'] if k['kernel'].startswith('lstm')])"
done
FC_LSTM_ROLES=1 timeout 600 python bench.py --workload freqcodec_gr1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r6/lstm/freq_roles1.json 2> gpurun_out/r6/lstm/freq_roles1.err
FC_LSTM_ROLES=0 timeout 600 python bench.py --workload freqcodec_gr1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r6/lstm/freq_roles0.json 2> gpurun_out/r6/lstm/freq_roles0.err
python - <<'P'
import json
for r in (1, 0):
    try:
        d = json.load(open(f"gpurun_out/r6/lstm/freq_roles{r}.json"))
        print("freqcodec roles", r, d.get("ms_per_step"), [(k["kernel"][:24], k["avg_us_per_launch"]) for k in d.get("kernels", []) if k["kernel"].startswith("lstm")])
    except Exception as ex:
        print("freq", r, ex)
P
