#!/bin/bash
mkdir -p gpurun_out/r5
for env in "FC_XQ=1" "FC_XQ=0"; do
  for spec in "ds320 8000" "ds640 6400"; do
    echo "=== $env $spec"
    env $env timeout 600 python tools/r5_layers_debug.py $spec 2>&1 | grep -v amdgpu.ids | grep "BAD\|Error\|error" | head -40
  done
done > gpurun_out/r5/layers_debug.txt 2>&1
cat gpurun_out/r5/layers_debug.txt
