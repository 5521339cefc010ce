#!/bin/bash
# A/B of library builds on the headline step in ONE call (boxes differ by ~3 %): usage r5_gpu_ab.sh default <variant> ...  (variant -> funcodec_amd/libfc_<variant>.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r5
cd $R
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" != "default" ]; then export FC_LIB=$R/funcodec_amd/libfc_$v.so; else unset FC_LIB; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r5/ab_$v.json 2> gpurun_out/r5/ab_$v.err
  python - <<P
import json
try:
    d = json.loads(open('gpurun_out/r5/ab_$v.json').read().strip().splitlines()[-1])
    print('$v rep $rep', d['ms_per_step'])
    if $rep == 2:
        for k in sorted(d.get('kernels', []), key=lambda k: -k['ms_per_step'])[:16]:
            print('   %-60s n=%d %.3f ms f32 %.3f' % (k['kernel'][:60], k['launches_per_step'], k['ms_per_step'], k['f32_frac'] or 0))
except Exception as e:
    print('$v failed', e); print(open('gpurun_out/r5/ab_$v.err').read()[-800:])
P
done
done
