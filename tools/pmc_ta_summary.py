"""Per-kernel summary of tools/pmc_ta.sh: how busy the texture-address units are.  TA_TA_BUSY_sum is summed over the chip's TA instances
(one per CU, 256) and GRBM_GUI_ACTIVE over the 8 XCDs, so TA busy % = TA_TA_BUSY_sum / 256 / (GRBM_GUI_ACTIVE / 8)."""
import sys
from pmc_laura_summary import load

acc, n, dur = load(sys.argv[1])
print(f"{'kernel':60s} {'launches':>8s} {'us/launch':>10s} {'TA busy %':>9s}")
for k, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]])[:22]:
    m = n[k]
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    pct = lambda name: 100.0 * c.get(name, 0.0) / 256.0 / cyc if cyc else 0.0
    print(f"{k[:60]:60s} {m:8d} {dur[k] / m:10.2f} {pct('TA_TA_BUSY_sum'):9.1f}")
