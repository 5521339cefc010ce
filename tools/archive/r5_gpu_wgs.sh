#!/bin/bash
# round 5: does the second workgroup per CU still pay for the DMA-staged conv classes?  FC_TARGET_WGS (always live) = workgroups a conv launch aims for
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r5
for w in 512 256 384 512; do
  FC_TARGET_WGS=$w timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r5/wgs_$w.json 2> gpurun_out/r5/wgs_$w.err
  python - <<P
import json
try:
    d = json.loads(open('gpurun_out/r5/wgs_$w.json').read().strip().splitlines()[-1])
    print('target_wgs=$w', d['ms_per_step'])
    for k in sorted(d.get('kernels', []), key=lambda k: -k['ms_per_step'])[:14]:
        if 'conv' in k['kernel']: print('   %-60s n=%d %.3f ms f32 %.3f' % (k['kernel'][:60], k['launches_per_step'], k['ms_per_step'], k['f32_frac'] or 0))
except Exception as e:
    print('target_wgs=$w failed', e); print(open('gpurun_out/r5/wgs_$w.err').read()[-600:])
P
done
