"""Post-process tools/collect_profiles.sh output: per-kernel HBM traffic of ONE benchmark step from the PMC passes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def per_kernel(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    # steps are delimited by the volume kernel (first kernel of every encode): keep the last step
    starts = [i for i, r in enumerate(rows) if "volume_kernel" in r["Kernel_Name"]]
    rows = rows[starts[-1]:] if starts else rows
    acc, n = defaultdict(float), defaultdict(int)
    for r in rows:
        nm = r["Kernel_Name"].replace("void fc::", "").replace("fc::", "").split("(")[0]
        acc[nm] += float(r["Counter_Value"])
        n[nm] += 1
    return acc, n


# optional second argument: a tag -- "freqcodec" reads pmc_fetch_freqcodec / pmc_write_freqcodec (the same two passes over
# `bench.py --workload freqcodec_gr1`) and writes hbm_traffic_pmc_freqcodec.json (one step = one 32-utterance engine call)
tag = ("_" + sys.argv[2]) if len(sys.argv) > 2 else ""
f = glob.glob(os.path.join(out, "pmc_fetch" + tag, "*counter_collection.csv"))
w = glob.glob(os.path.join(out, "pmc_write" + tag, "*counter_collection.csv"))
res = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes, last benchmark step ("
               + ("FreqCodec gr1, one 32 x 10 s engine call" if tag else "ds640, 16 x 10 s") + "); "
               "raw counter units are KB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) "
               "coalesced reads -> fetch_gb_x2 doubles it; dword reads and WRITE_SIZE are uncalibrated there, so both raw and "
               "doubled read figures are given.", "per_kernel": []}
if f and w:
    fa, fn = per_kernel(f[0], "FETCH_SIZE")
    wa, _ = per_kernel(w[0], "WRITE_SIZE")
    tot_f = tot_w = 0.0
    for k in sorted(fa, key=lambda k: -(fa[k] + wa.get(k, 0.0))):
        res["per_kernel"].append({"kernel": k, "launches": fn[k], "fetch_kb": fa[k], "write_kb": wa.get(k, 0.0)})
        tot_f += fa[k]
        tot_w += wa.get(k, 0.0)
    res["step_total"] = {"fetch_gb_raw": tot_f / 1e6, "fetch_gb_x2": 2 * tot_f / 1e6, "write_gb": tot_w / 1e6}
json.dump(res, open(os.path.join(out, "hbm_traffic_pmc" + tag + ".json"), "w"), indent=1)
st = glob.glob(os.path.join(out, "rp", "*kernel_stats.csv"))
if st and not tag:
    open(os.path.join(out, "kernel_stats.csv"), "w").write(open(st[0]).read())
print(json.dumps(res.get("step_total", {})))
