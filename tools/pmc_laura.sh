#!/bin/bash
# PMC evidence for the LauraTTS side measurement (BASELINE.json configs[4]; run through gpurun from the repo root):
#   pass 1  MFMA-pipe / busy / clock counters      pass 2  FETCH_SIZE      pass 3  WRITE_SIZE   (separate passes, kernel-trace only)
# Output: gpurun_out/pmc_laura/{sq,fetch,write}/*.csv, summarised by tools/pmc_laura_summary.py into profiles/rNN_pmc_laura.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_laura
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload laura --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU -d $OUT/sq -o out --output-format csv -- $CMD > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o out --output-format csv -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o out --output-format csv -- $CMD > /dev/null 2> $OUT/write.err
python $R/tools/pmc_laura_summary.py $OUT > $OUT/summary.txt 2>&1
# keep the merge-back small
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT $OUT/sq | head -20
