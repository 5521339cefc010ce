#!/bin/bash
# round 6, call 1: (a) parity subset + headline bench of the build whose ablation switches are compile-time (FC_ABL), (b) what the fused
# prologue ARITHMETIC costs per conv class: FC_ABLATE 256 (staging waves store the loaded values unchanged) and 512 (no slab write at all)
# of an FC_AB_KNOBS build against mask 0 of the same build (one process per mask, same box, same call)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r6/abl
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_conv_layer or padding_edge or staging_scheme or fused_resblock or e2e_against_reference_golden or lstm" > gpurun_out/r6/pytest_1.log 2>&1
tail -3 gpurun_out/r6/pytest_1.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r6/bench_hyg.json 2> gpurun_out/r6/bench_hyg.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r6/bench_hyg.json"))
print("default build ms_per_step", d["ms_per_step"], "conv", d["roofline"].get("conv_class", {}).get("all_conv_instantiations"))
P
export FC_LIB=$R/funcodec_amd/libfc_ab.so
for m in 0 256 512; do
  FC_ABLATE=$m timeout 200 python tools/ablate_step.py gpurun_out/r6/abl/mask_$m.json 2>&1 | grep -v amdgpu.ids | tail -1
done
python - <<'P' | tee gpurun_out/r6/prologue_ablation.txt
import json, os
masks = [0, 256, 512]
data = {m: json.load(open(f"gpurun_out/r6/abl/mask_{m}.json"))["classes"] for m in masks if os.path.exists(f"gpurun_out/r6/abl/mask_{m}.json")}
base = data[0]
print("%-58s %2s %8s %6s %10s %10s" % ("class", "n", "us", "frac", "noPrologue", "noSlabWrite"))
tot = {m: 0.0 for m in data}
for k, v in sorted(base.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if not k.startswith(("conv", "reshead")):
        continue
    line = "%-58s %2d %8.1f %6.3f" % (k[:58], v["launches_per_step"], v["us_per_launch"], v["tflops"] / 157.3)
    for m in masks[1:]:
        line += " %10s" % ("%.1f" % data[m][k]["us_per_launch"] if m in data and k in data[m] else "-")
    for m in data:
        if k in data[m]:
            tot[m] += data[m][k]["ms_per_step"]
    print(line)
print("all conv classes ms per step", {m: round(t, 3) for m, t in tot.items()})
P
