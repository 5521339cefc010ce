"""Summarise a rocprofv3 kernel trace CSV: one line per launch of the last benchmark step."""
import csv
import sys


def main(path, nsteps):
    rows = list(csv.DictReader(open(path)))
    # find step boundaries by the volume kernel (first kernel of every encode)
    starts = [i for i, r in enumerate(rows) if "volume_kernel" in r["Kernel_Name"]]
    lo = starts[-1]
    hi = len(rows)
    out = []
    gaps = []                                          # (gap before this launch in us, kernel, previous kernel)
    prev_end, prev_nm = None, ""
    for r in sorted(rows[lo:hi], key=lambda r: int(r["Start_Timestamp"])):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if prev_end is not None:
            gaps.append(((int(r["Start_Timestamp"]) - prev_end) / 1e3, r["Kernel_Name"].replace("void fc::", "").split("(")[0][:40], prev_nm))
        prev_end = max(prev_end or 0, int(r["End_Timestamp"]))
        prev_nm = r["Kernel_Name"].replace("void fc::", "").split("(")[0][:40]
        nm = r["Kernel_Name"].replace("void fc::", "").replace("fc::", "").split("(")[0]
        g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        if out and out[-1][0] == nm and "lstm" in nm:
            out[-1][1] += d
            out[-1][3] += 1
        else:
            out.append([nm, d, g, 1, r["LDS_Block_Size"], r["VGPR_Count"], r.get("Accum_VGPR_Count", "")])
    t0 = int(rows[lo]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in rows[lo:hi])
    print("span us", (t1 - t0) / 1e3, "sum kernel us", sum(o[1] for o in out))
    agg = {}
    for o in out:
        agg[o[0]] = agg.get(o[0], 0) + o[1]
        if "gn_finalize" in o[0]:
            continue
        print(f"{o[0][:44]:44s} {o[1]:9.1f} us grid={o[2]} n={o[3]} lds={o[4]} vgpr={o[5]}+{o[6]}")
    pos = [g for g in gaps if g[0] > 0]
    print("--- gaps between consecutive launches: n=%d sum=%.1f us, median %.2f us, > 10 us: %d" % (
        len(pos), sum(g[0] for g in pos), sorted(g[0] for g in pos)[len(pos) // 2] if pos else 0.0, sum(1 for g in pos if g[0] > 10)))
    for g in sorted(gaps, key=lambda g: -g[0])[:8]:
        print("   %7.1f us before %-40s (after %s)" % g)
    print("--- totals")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
        print(f"{k[:60]:60s} {v:10.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
