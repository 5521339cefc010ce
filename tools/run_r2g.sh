#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r2g_test.log
tail -4 gpurun_out/r2g_test.log
python tools/ablate_rvq.py 2>&1 | grep ablate
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-event-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=16 ms_per_step', d['ms_per_step'], d['value'])"
FC_BENCH_UTTS=128 FC_BENCH_MICRO=16 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-event-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('128 utts micro 16: ms_per_step', d['ms_per_step'], d['value'], d['ms_per_micro_batch'])"
FC_BENCH_UTTS=128 FC_BENCH_MICRO=32 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-event-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('128 utts micro 32: ms_per_step', d['ms_per_step'], d['value'], d['ms_per_micro_batch'])"
