#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench.json               the official bench line (with cpu_baseline)
#   bench_under_rocprof.json the same command under rocprofv3 --kernel-trace --stats
#   kernel_stats.csv         rocprofv3 per-kernel statistics of that run
#   last_step_per_launch.txt one line per launch of the last benchmark step (tools/trace_summary.py)
#   pmc_fetch / pmc_write    FETCH_SIZE and WRITE_SIZE in SEPARATE passes (kernel-trace only), as the guide prescribes
# Outputs go to gpurun_out/prof/; tools/profile_post.py turns them into profiles/rNN_*.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/rp -o out --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/rp.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o out --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-event-profile --no-secondary > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o out --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-event-profile --no-secondary > /dev/null 2> $OUT/pmc_write.err
python $R/tools/trace_summary.py $(ls $OUT/rp/*kernel_trace.csv | head -1) > $OUT/last_step_per_launch.txt 2>&1
python $R/tools/profile_post.py $OUT
# side measurements of the next scope rows on their own (BASELINE.json configs[3] shape with grouped convs, configs[4])
python $R/bench.py --workload freqcodec_gr1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_freqcodec_gr1.json 2> $OUT/bench_freqcodec_gr1.err
rocprofv3 --kernel-trace --stats -d $OUT/rpf -o out --output-format csv -- python $R/bench.py --workload freqcodec_gr1 --steps 2 --warmup 1 > /dev/null 2> $OUT/rpf.err
cp $(ls $OUT/rpf/*kernel_stats.csv | head -1) $OUT/kernel_stats_freqcodec_gr1.csv
rm -rf $OUT/rpf
# FreqCodec HBM traffic (round 6): the same two separate PMC passes over the gr1 side measurement -> hbm_traffic_pmc_freqcodec.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_freqcodec -o out --output-format csv -- python $R/bench.py --workload freqcodec_gr1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch_freqcodec.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_freqcodec -o out --output-format csv -- python $R/bench.py --workload freqcodec_gr1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write_freqcodec.err
python $R/tools/profile_post.py $OUT freqcodec
rm -rf $OUT/pmc_fetch_freqcodec $OUT/pmc_write_freqcodec
python $R/bench.py --workload laura --steps 5 --warmup 2 > $OUT/bench_laura.json 2> $OUT/bench_laura.err
rocprofv3 --kernel-trace --stats -d $OUT/rpl -o out --output-format csv -- python $R/bench.py --workload laura --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/rpl.err
cp $(ls $OUT/rpl/*kernel_stats.csv | head -1) $OUT/kernel_stats_laura.csv
rm -rf $OUT/rpl
ls -la $OUT | head -30
# keep the merge-back small: raw traces are large
rm -rf $OUT/rp/*kernel_trace.csv $OUT/pmc_fetch $OUT/pmc_write
