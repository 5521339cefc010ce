#!/bin/bash
for extra in 0 8192; do
echo "== FC_LDS_EXTRA=$extra shortcut T=100"; FC_LDS_EXTRA=$extra python tools/r5_partials_probe.py encoder.model.13.shortcut.conv 100 1 128 128 1 1 0 0 2>&1 | grep -v amdgpu
done
