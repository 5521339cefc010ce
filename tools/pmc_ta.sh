#!/bin/bash
# Address-path (TA) pressure per kernel: which kernels keep the texture-address unit busy / stalled (the round-3 LSTM finding: 16-byte
# loads whose lanes sit on different cache lines are bound here, not by bandwidth or latency).  One rocprofv3 --pmc pass per workload
# (kernel-trace only), summarised by tools/pmc_ta_summary.py.  usage (through gpurun, from the repo root): bash tools/pmc_ta.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ta
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in encodec freqcodec_gr1 laura; do
  CMD="python $R/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-event-profile"
  # two counters per pass (more TA / TCP counters at once: "Request exceeds the capabilities of the hardware", after which rocprofv3 does
  # not exit on its own -- hence the timeout)
  timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum -d $OUT/$w -o out --output-format csv -- $CMD > /dev/null 2> $OUT/$w.err
  python $R/tools/pmc_ta_summary.py $OUT/$w > $OUT/$w.txt 2>&1
  find $OUT/$w -name "*kernel_trace.csv" -delete
done
tail -n +1 $OUT/*.txt | head -120
