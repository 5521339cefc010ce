"""Tuning aid: summarise the s_memtime stamps of one persistent LauraTTS decoding step (FC_LAURA_TRACE=<file> python bench.py --workload laura).
Per phase kind: mean / max of wait, staging, MFMA + reduction, epilogue + arrival (shader clocks -> us at the given GHz)."""
import sys

import numpy as np

path, ghz = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2.1
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 12
t = np.fromfile(path, dtype=np.uint64).reshape(256, 64, 8).astype(np.int64)
kinds = ["QKV", "ATT", "OUT", "FF1", "FF2", "DEC"]
rows = {k: [] for k in kinds}
per_phase = {}
for wg in range(256):
    for i in range(64):
        s = t[wg, i]
        if s[0] == 0:
            continue
        ph, u = int(s[7] >> 32), int(s[7] & 0xffffffff)
        k = 5 if ph == 5 * nl else ph % 5
        wait = s[1] - s[0]
        if k == 1:
            stage, mfma, epi = 0, s[3] - s[1], s[4] - s[3]
        else:
            stage, mfma, epi = s[2] - s[1], s[3] - s[2], s[4] - s[3]
        extra = (s[5] - s[1], s[6] - s[5], s[2] - s[6], s[3] - s[2]) if k != 1 else (0, 0, 0, 0)
        rows[kinds[k]].append((wait, stage, mfma, epi) + tuple(extra))
        per_phase.setdefault(ph, []).append((s[0], s[1], s[4], wg))
print(f"{'kind':4s} {'units':>6s} | wait mean/max | stage mean/max | compute mean/max | epilogue+arrive mean/max   (us at {ghz} GHz)")
for k in kinds:
    a = np.array(rows[k], dtype=np.float64) / (ghz * 1e3)
    if not len(a):
        continue
    print(f"{k:4s} {len(a):6d} | " + " | ".join(f"{a[:, j].mean():6.2f} {a[:, j].max():6.2f}" for j in range(4)) +
          "  || loads landed %.2f, first barrier %.2f, LN + barrier %.2f, mfma + dma issue + red barrier %.2f" % tuple(a[:, 4 + j].mean() for j in range(4)))
# per-workgroup view of a step: first stamp to last stamp
span = []
for wg in range(256):
    v = t[wg][t[wg, :, 0] > 0]
    if len(v):
        span.append((v[:, 4].max() - v[:, 0].min()) / (ghz * 1e3))
print(f"per-workgroup span of the step: mean {np.mean(span):.1f} us, max {np.max(span):.1f} us over {len(span)} workgroups")
# within one workgroup (its own clock): time from the end of a unit to the end of the wait of its next unit, by phase kind of the next unit
