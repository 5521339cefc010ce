"""Summarise rocprofv3 --pmc counter_collection CSVs for the conv kernel: mean per launch of every counter."""
import csv
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    acc = defaultdict(list)
    dur = []
    seen = set()
    names = set()
    for r in csv.DictReader(open(path)):
        if "conv_mfma_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        names.add(r["Kernel_Name"].split("(")[0].replace("void fc::", ""))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    n = len(dur)
    print(path.split("/")[-2], f"launches={n} mean_us={sum(dur[3:]) / max(1, n - 3):.1f}", " ".join(sorted(names)))
    for k, v in sorted(acc.items()):
        print(f"   {k:28s} {sum(v[3:]) / max(1, len(v) - 3):16.0f}")
