#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -m gpu -q -x -k "resblock or golden or full_size" 2>&1 | tail -15 > gpurun_out/r2d_test.log
tail -4 gpurun_out/r2d_test.log
for w in 512 768 1024; do FC_RH_WGS=$w python tools/ablate_reshead.py encoder.model.1 160000 2>&1 | grep reshead; done
python tools/ablate_reshead.py encoder.model.4 80000 2>&1 | grep reshead
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-event-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
