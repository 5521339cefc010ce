"""Per-kernel means of the rocprofv3 --pmc passes of tools/pmc_laura.sh.
MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
(MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs); HBM bytes = FETCH_SIZE x 2 (gfx950 correction of the guide, checked in
round 1 on layers of known size) x 1024 + WRITE_SIZE x 1024."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    dur = defaultdict(float)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in cnt[k]:
                cnt[k].add(r["Dispatch_Id"])
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc, {k: len(v) for k, v in cnt.items()}, dur


def main():
    root = sys.argv[1]
    sq, n, dur = load(os.path.join(root, "sq"))
    fe, nf, _ = load(os.path.join(root, "fetch"))
    wr, nw, _ = load(os.path.join(root, "write"))
    rows = sorted(sq.items(), key=lambda kv: -dur[kv[0]])
    print(f"{'kernel':58s} {'launches':>8s} {'us/launch':>10s} {'GHz':>6s} {'MFMA busy %':>11s} {'MFMA insts':>11s} {'fetch MB':>9s} {'write MB':>9s}  (per launch)")
    for k, c in rows[:24]:
        m = n[k]
        cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / m
        us = dur[k] / m
        busy = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / m / (1024.0 * cycles) if cycles else 0.0
        f = 2.0 * fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 / max(1, nf.get(k, 1)) / 1e6
        w = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / max(1, nw.get(k, 1)) / 1e6
        print(f"{k[:58]:58s} {m:8d} {us:10.2f} {cycles / us / 1e3 if us else 0:6.2f} {busy:11.1f} {c.get('SQ_INSTS_MFMA', 0.0) / m:11.0f} {f:9.3f} {w:9.3f}")


if __name__ == "__main__":
    main()
