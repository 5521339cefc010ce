#!/bin/bash
# NOTE: FC_ROW_NSET / FC_ELEM_NSET are BUILD defines now (the nset1 library of this script = the default build; "default" here was FC_ROW_NSET=2 FC_ELEM_NSET=2).
# round 6, call 2: two register sets in the quad staging paths (default build) against the one-set form (libfc_nset1.so): parity subset on the
# default build, per-class tables and the headline bench of both in the same call; FreqCodec PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r6/ab2 gpurun_out/prof
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_conv_layer or padding_edge or random_shape_sweep or staging_scheme or fused_resblock or e2e_against_reference_golden or segmented or workspace_contents" > gpurun_out/r6/pytest_2.log 2>&1
tail -3 gpurun_out/r6/pytest_2.log
for v in default nset1 default nset1; do
  if [ $v = default ]; then unset FC_LIB; else export FC_LIB=$R/funcodec_amd/libfc_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r6/ab2/bench_$v.json 2> gpurun_out/r6/ab2/bench_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r6/ab2/bench_$v.json')); print('$v', 'ms_per_step', d['ms_per_step'])"
done
for v in default nset1; do
  if [ $v = default ]; then unset FC_LIB; else export FC_LIB=$R/funcodec_amd/libfc_$v.so; fi
  timeout 200 python tools/ablate_step.py gpurun_out/r6/ab2/classes_$v.json 2>&1 | grep -v amdgpu.ids | tail -1
done
unset FC_LIB
python - <<'P' | tee gpurun_out/r6/ab2/classes.txt
import json
a = json.load(open("gpurun_out/r6/ab2/classes_nset1.json"))["classes"]
b = json.load(open("gpurun_out/r6/ab2/classes_default.json"))["classes"]
print("%-58s %2s %9s %9s" % ("class", "n", "one set", "two sets"))
for k, v in sorted(a.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if k in b:
        print("%-58s %2d %9.1f %9.1f" % (k[:58], v["launches_per_step"], v["us_per_launch"], b[k]["us_per_launch"]))
P
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_freqcodec -o out --output-format csv -- python $R/bench.py --workload freqcodec_gr1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch_freqcodec.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_freqcodec -o out --output-format csv -- python $R/bench.py --workload freqcodec_gr1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write_freqcodec.err
python $R/tools/profile_post.py $OUT freqcodec
ls -la $OUT | head; rm -rf $OUT/pmc_fetch_freqcodec $OUT/pmc_write_freqcodec
