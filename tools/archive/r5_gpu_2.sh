#!/bin/bash
# round 5, call 2: quad-k conv kernel -- layer-level parity first, then the suite's conv / e2e subset, then a short bench A/B against FC_QUAD=0
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_conv_layer or padding_edge or random_shape_sweep or row_staging or staging_scheme or fused_resblock" > gpurun_out/r5/pytest_conv.log 2>&1
tail -15 gpurun_out/r5/pytest_conv.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r5/bench_quad.json 2> gpurun_out/r5/bench_quad.err
FC_QUAD=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r5/bench_noquad.json 2> gpurun_out/r5/bench_noquad.err
python - <<'P'
import json
for n in ("quad","noquad"):
    try:
        d=json.loads(open(f"gpurun_out/r5/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"])
        for k in sorted(d.get("kernels",[]), key=lambda k:-k["ms_per_step"])[:30]:
            print("   %-62s n=%d %.3f ms  %.1f us  f32 %.3f"%(k["kernel"][:62],k["launches_per_step"],k["ms_per_step"],k["avg_us_per_launch"],k["f32_frac"] or 0))
    except Exception as e:
        print(n,"failed",e); print(open(f"gpurun_out/r5/bench_{n}.err").read()[-1500:])
P
