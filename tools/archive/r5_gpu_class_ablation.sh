#!/bin/bash
# round 5 (VERDICT r4 #2b): per conv class of the headline step, the launch time as built and without the MFMAs / the input-slab staging /
# the weight DMA / the stores / the epilogue (FC_ABLATE masks of an FC_AB_KNOBS build; one process per mask, same box, same call)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r5/abl
export FC_LIB=$R/funcodec_amd/libfc_ab.so
for m in 0 1 4 16 20 2 128 21; do
  FC_ABLATE=$m timeout 200 python tools/ablate_step.py gpurun_out/r5/abl/mask_$m.json 2>&1 | grep -v amdgpu.ids | tail -2
done
python - <<'P' > gpurun_out/r5/conv_class_ablation.txt
import json, glob, os
masks = [0, 1, 4, 16, 20, 2, 128, 21]
names = {0: "as built", 1: "no MFMA", 4: "no slab staging", 16: "no weight DMA", 20: "no staging at all", 2: "no stores", 128: "no epilogue", 21: "no MFMA, no staging"}
data = {}
for m in masks:
    p = f"gpurun_out/r5/abl/mask_{m}.json"
    if os.path.exists(p):
        data[m] = json.load(open(p))["classes"]
base = data[0]
print("Headline step (ds640, 16 x 10 s, n_q = 32): us per launch of every conv class, as built and under FC_ABLATE masks (tools/r5_gpu_class_ablation.sh;")
print("one process per mask on the same box; results under a mask are garbage by design).  frac = fraction of the 157.3 TFLOP/s fp32 matrix peak as built,")
print("TB/s = algorithmic HBM bytes of the launch (inputs + weights + outputs once) / its time, against ~8 TB/s.")
print("Reading: 'no MFMA' is what the staging + epilogue cost when nothing competes for issue slots; 'no staging at all' is the matrix loop + epilogue alone;")
print("the as-built time is close to their SUM, not their maximum, for the classes below 0.60 (fp32 MFMA and vector issue share the SIMD).")
print()
hdr = "%-58s %2s %7s %5s %5s" % ("class", "n", "us", "frac", "TB/s") + "".join(" %12s" % names[m][:12] for m in masks[1:] if m in data)
print(hdr)
for k, v in sorted(base.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if not k.startswith(("conv", "reshead")):
        continue
    line = "%-58s %2d %7.1f %5.3f %5.2f" % (k[:58], v["launches_per_step"], v["us_per_launch"], v["tflops"] / 157.3, v.get("alg_tbs", 0.0))
    for m in masks[1:]:
        if m in data:
            line += " %12s" % ("%.1f" % data[m][k]["us_per_launch"] if k in data[m] else "-")
    print(line)
P
cat gpurun_out/r5/conv_class_ablation.txt | cut -c1-200
