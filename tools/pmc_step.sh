#!/bin/bash
# SQ counters of every kernel class of the headline step (two --pmc passes, kernel-trace only, as the guide prescribes) -> gpurun_out/pmc_step/summary.txt
# (committed as profiles/rNN_pmc_step.txt): what the waves of each conv class spend their cycles on (MFMA busy, issue stalls, parked, LDS conflicts)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_step
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-event-profile --no-secondary"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $OUT/a -o out --output-format csv -- $CMD > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU -d $OUT/b -o out --output-format csv -- $CMD > /dev/null 2> $OUT/b.err
python $R/tools/pmc_step_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
cat $OUT/summary.txt | cut -c1-230
