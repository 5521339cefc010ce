"""Per kernel class of the LAST headline step: SQ counters from the two passes of tools/pmc_step.sh.
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts
cycles summed over SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Columns:
  mfma%   SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8
  parked% SQ_WAIT_ANY / SQ_WAVE_CYCLES        (s_waitcnt / s_barrier: waves that cannot issue because they wait for memory, LDS or a barrier)
  stall%  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (issue stalls: MFMA read-after-write / pipe busy);  of which LDS-issue stall: SQ_WAIT_INST_LDS
  act%    SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (cycles in which a wave issues something)
  ldsconf SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS cycles over all LDS-array cycles)
  valu/mfma, lds/mfma  instruction ratios (SQ_INSTS_VALU excludes MFMA)
  GHz     GRBM_GUI_ACTIVE / 8 / kernel time: the shader clock the chip sustained inside the class (the 157.3 TFLOP/s roof assumes 2.4)"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    dur = defaultdict(float)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = [r for f in files for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "volume_kernel" in r["Kernel_Name"] and r["Counter_Name"] == rows[0]["Counter_Name"]]
    lo = starts[-1] if starts else 0
    first = int(rows[lo]["Start_Timestamp"])
    for r in rows:
        if int(r["Start_Timestamp"]) < first:
            continue
        k = r["Kernel_Name"].replace("void fc::", "").replace("fc::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]:
            n[k].add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc, {k: len(v) for k, v in n.items()}, dur


def main():
    root = sys.argv[1]
    a, na, dur = load(os.path.join(root, "a"))
    b, nb, _ = load(os.path.join(root, "b"))
    print(__doc__)
    print("%-56s %3s %8s %6s %7s %6s %6s %5s %7s %9s %8s %5s" % ("kernel class (last step)", "n", "us/launch", "mfma%", "parked%", "stall%", "ldsst%", "act%", "ldsconf", "valu/mfma", "lds/mfma", "GHz"))
    for k in sorted(a, key=lambda k: -dur[k]):
        c, d = a[k], b.get(k, {})
        wave = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        kc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 or 1.0
        mf = d.get("SQ_INSTS_MFMA", 0.0)
        print("%-56s %3d %8.1f %6.1f %7.1f %6.1f %6.1f %5.1f %7.3f %9s %8s %5.2f" % (
            k[:56], na[k], dur[k] / na[k], 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * kc),
            100.0 * c.get("SQ_WAIT_ANY", 0.0) / wave, 100.0 * c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 100.0 * c.get("SQ_WAIT_INST_LDS", 0.0) / wave,
            100.0 * c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave,
            d.get("SQ_LDS_BANK_CONFLICT", 0.0) / (d.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0),
            ("%.2f" % (d.get("SQ_INSTS_VALU", 0.0) / mf)) if mf else "-", ("%.2f" % (d.get("SQ_INSTS_LDS", 0.0) / mf)) if mf else "-",
            kc / (dur[k] * 1e3) if dur[k] else 0.0))


if __name__ == "__main__":
    main()
