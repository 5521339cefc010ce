#!/bin/bash
for env in "FC_XQ=1" "FC_XQ=0" "FC_QUAD=0"; do
  for spec in "32 160000" "2 160000"; do echo "== $env $spec"; env $env timeout 300 python tools/r5_freq_nan.py $spec 2>&1 | grep -v amdgpu | tail -3; done
done
