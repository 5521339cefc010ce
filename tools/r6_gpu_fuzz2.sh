#!/bin/bash
# round 6, final build: wide sweeps (time-domain / FreqCodec architectures, LauraTTS) and repeated-call determinism
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/r6
mkdir -p $OUT
timeout 1500 python tools/fuzz_archs.py 8000 8500 > $OUT/fuzz_time_wide.txt 2>&1; tail -1 $OUT/fuzz_time_wide.txt; grep -c "frames_bad" $OUT/fuzz_time_wide.txt
FREQ=1 timeout 1500 python tools/fuzz_archs.py 8000 8160 > $OUT/fuzz_freq_wide.txt 2>&1; tail -1 $OUT/fuzz_freq_wide.txt; grep -c "frames_bad" $OUT/fuzz_freq_wide.txt
timeout 900 python tools/fuzz_laura.py 60 900 > $OUT/fuzz_laura_wide.txt 2>&1; tail -1 $OUT/fuzz_laura_wide.txt
timeout 600 python tools/stress_determinism.py > $OUT/stress_determinism.txt 2>&1; tail -2 $OUT/stress_determinism.txt
