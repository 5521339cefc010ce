#!/bin/bash
# round 6, final build: pseudo-random architecture / length / LauraTTS sweeps beyond the suite's seeds (the workgroup-count selection touches
# every conv launch of every recipe)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/r6
mkdir -p $OUT
timeout 900 python tools/fuzz_archs.py 6000 6200 > $OUT/fuzz_time.txt 2>&1; tail -1 $OUT/fuzz_time.txt; grep -c "frames_bad" $OUT/fuzz_time.txt; grep CHECK $OUT/fuzz_time.txt | head -5
FREQ=1 timeout 900 python tools/fuzz_archs.py 6000 6080 > $OUT/fuzz_freq.txt 2>&1; tail -1 $OUT/fuzz_freq.txt; grep -c "frames_bad" $OUT/fuzz_freq.txt; grep CHECK $OUT/fuzz_freq.txt | head -5
timeout 900 python tools/fuzz_lengths.py > $OUT/fuzz_lengths.txt 2>&1; tail -2 $OUT/fuzz_lengths.txt
timeout 600 python tools/fuzz_laura.py 30 700 > $OUT/fuzz_laura.txt 2>&1; tail -2 $OUT/fuzz_laura.txt
