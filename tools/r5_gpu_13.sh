#!/bin/bash
mkdir -p gpurun_out/r5
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r5/pytest_full.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r5/pytest_full.log | head -60
