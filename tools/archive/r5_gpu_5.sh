#!/bin/bash
# round 5, call 5: DMA-staged quad slabs (MODE 5): layer parity, e2e goldens, bench A/B (xq+quad | quad only | round-4 layout), timelines
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_conv_layer or padding_edge or random_shape_sweep or row_staging or staging_scheme or fused_resblock or e2e_against_reference_golden or lstm_against" > gpurun_out/r5/pytest_conv5.log 2>&1
tail -8 gpurun_out/r5/pytest_conv5.log
run() { n=$1; shift; env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r5/bench_$n.json 2> gpurun_out/r5/bench_$n.err; }
run xq FC_XQ=1
run quad FC_XQ=0
run r4 FC_QUAD=0
python - <<'P'
import json
for n in ("xq","quad","r4"):
    try:
        d=json.loads(open(f"gpurun_out/r5/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"])
        for k in sorted(d.get("kernels",[]), key=lambda k:-k["ms_per_step"])[:24]:
            print("   %-62s n=%d %.3f ms  %.1f us  f32 %.3f"%(k["kernel"][:62],k["launches_per_step"],k["ms_per_step"],k["avg_us_per_launch"],k["f32_frac"] or 0))
    except Exception as e:
        print(n,"failed",e); print(open(f"gpurun_out/r5/bench_{n}.err").read()[-1500:])
P
export FC_LIB=$PWD/funcodec_amd/libfuncodec_amd_timeline.so
for spec in "decoder.model.9.convtr 10000 elu" "encoder.model.15.conv 2000 elu" "decoder.model.3.convtr 250 elu"; do
    echo "=== xq $spec"
    timeout 300 python tools/ablate_layer.py $spec 2>&1 | grep -v "amdgpu.ids" | awk 'NR<=3 || /item  [5-9] |item 1[0-4] /'
done > gpurun_out/r5/timeline_xq.txt 2>&1
cat gpurun_out/r5/timeline_xq.txt
