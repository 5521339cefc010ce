"""Per-kernel statistics of a rocprofv3 run stored as a rocpd SQLite database (ROCm 7's default output):
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--csv out.csv] [--top 30] [--gaps]"""
import re
import sqlite3
import sys


def demangle(names):
    import shutil
    import subprocess
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([tool], input="\n".join(n.replace(".kd", "") for n in names), capture_output=True, text=True, check=True).stdout
        res = out.strip("\n").split("\n")
        return res if len(res) == len(names) else list(names)
    except Exception:
        return list(names)


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 30
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute("""select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                          from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                          group by s.kernel_name order by 3 desc""").fetchall()
    dn = demangle([r[0] for r in rows])
    rows = [(re.sub(r"^void ", "", re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", d)),) + tuple(r[1:]) for d, r in zip(dn, rows)]
    tot = sum(r[2] for r in rows)
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} launches")
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for r in rows:
        name = r[0].replace(".kd", "")
        lines.append(f'"{name}",{r[1]},{r[2]},{r[3]:.1f},{100.0 * r[2] / tot:.2f},{r[4]},{r[5]}')
    for r in rows[:top]:
        name = re.sub(r"\(.*", "", r[0])[:72]
        print(f"{name:72s} n={r[1]:6d} tot={r[2] / 1e6:9.3f}ms {100.0 * r[2] / tot:5.1f}% avg={r[3] / 1e3:8.2f}us min={r[4] / 1e3:7.2f} max={r[5] / 1e3:8.2f}")
    if "--csv" in sys.argv:
        with open(sys.argv[sys.argv.index("--csv") + 1], "wt") as f:
            f.write("\n".join(lines) + "\n")
    if "--gaps" in sys.argv:      # idle time between consecutive dispatches (launch-bound loops)
        ev = cur.execute("select start, end from rocpd_kernel_dispatch order by start").fetchall()
        gaps = [b[0] - a[1] for a, b in zip(ev, ev[1:]) if 0 <= b[0] - a[1] < 200000]
        gaps.sort()
        if gaps:
            print(f"gaps between consecutive kernels: n={len(gaps)} median {gaps[len(gaps) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, "
                  f"p90 {gaps[int(len(gaps) * 0.9)] / 1e3:.2f} us, sum {sum(gaps) / 1e6:.2f} ms")


if __name__ == "__main__":
    main()
