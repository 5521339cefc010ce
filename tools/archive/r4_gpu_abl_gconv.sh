cd $GRAFT_REPO_ROOT
for v in default abl1 abl2 abl3; do
  if [ "$v" != "default" ]; then export FC_LIB=$PWD/funcodec_amd/libfc_$v.so; else unset FC_LIB; fi
  timeout 300 python bench.py --workload freqcodec_gr1 --steps 4 --warmup 2 > gpurun_out/bench_freq_$v.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_freq_$v.json"))
print("$v", d["ms_per_step"], [(round(k["ms_per_step"], 2), k["kernel"][:36]) for k in d["kernels"] if "gconv2d_kernel<4, 2, 3" in k["kernel"] or "8, 2, 1" in k["kernel"] or "<2, 2, 1, 1" in k["kernel"]])
PY
done
