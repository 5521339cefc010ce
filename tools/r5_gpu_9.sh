#!/bin/bash
for xq in 1 0; do
echo "== FC_XQ=$xq shortcut T=100"; FC_XQ=$xq python tools/r5_partials_probe.py encoder.model.13.shortcut.conv 100 1 128 128 1 1 0 0 2>&1 | grep -v amdgpu
done
echo "== FC_XQ=1 enc.12 T=3000"; FC_XQ=1 python tools/r5_partials_probe.py encoder.model.12.conv 3000 1 128 128 5 10 3 2 2>&1 | grep -v amdgpu | head -24
