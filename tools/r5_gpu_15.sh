#!/bin/bash
for env in "FC_XQ=1" "FC_XQ=0" "FC_QUAD=0"; do
  echo "== $env"; env $env timeout 300 python bench.py --workload freqcodec_gr1 --steps 3 --warmup 1 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400
done
