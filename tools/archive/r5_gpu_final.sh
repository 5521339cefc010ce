#!/bin/bash
# round 5, final build: the whole GPU suite + smoke + the default bench line, then the robustness sweeps (random 1-D / 2-D architectures,
# odd lengths) that exercise the quad-k / materialised-input conv paths outside the fixtures' shapes -- one gpurun call
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/r5_gpu_full.sh
OUT=$R/gpurun_out/r5
timeout 400 python tools/fuzz_archs.py 3000 3070 > $OUT/fuzz_time.txt 2>&1; tail -1 $OUT/fuzz_time.txt; grep -c CHECK $OUT/fuzz_time.txt; grep CHECK $OUT/fuzz_time.txt | head -5
FREQ=1 timeout 400 python tools/fuzz_archs.py 3000 3030 > $OUT/fuzz_freq.txt 2>&1; tail -1 $OUT/fuzz_freq.txt; grep CHECK $OUT/fuzz_freq.txt | head -5
timeout 400 python tools/fuzz_lengths.py > $OUT/fuzz_lengths.txt 2>&1; tail -2 $OUT/fuzz_lengths.txt
