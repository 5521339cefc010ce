#!/bin/bash
# round 5, final build: wider pseudo-random sweeps than the suite's seeds (time-domain architectures, 2-D FreqCodec architectures, LauraTTS)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/r5
mkdir -p $OUT
timeout 900 python tools/fuzz_archs.py 4000 4220 > $OUT/fuzz_time_wide.txt 2>&1; tail -1 $OUT/fuzz_time_wide.txt; grep -c "frames_bad" $OUT/fuzz_time_wide.txt; grep CHECK $OUT/fuzz_time_wide.txt | head -5
FREQ=1 timeout 900 python tools/fuzz_archs.py 4000 4090 > $OUT/fuzz_freq_wide.txt 2>&1; tail -1 $OUT/fuzz_freq_wide.txt; grep -c "frames_bad" $OUT/fuzz_freq_wide.txt; grep CHECK $OUT/fuzz_freq_wide.txt | head -5
[ -f tools/fuzz_laura.py ] && (timeout 600 python tools/fuzz_laura.py 40 500 > $OUT/fuzz_laura_wide.txt 2>&1; tail -2 $OUT/fuzz_laura_wide.txt)
