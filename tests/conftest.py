import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


# ---- tie waivers as a RECORD: every test that accepts a differing code index as a proven fp32 tie appends
# (case, stage, frame, margin, bound) here; the session prints one summary line (it lands in the tail of the driver's test log) and
# writes the list to $FC_WAIVER_JSON (default gpurun_out/tie_waivers.json, which gpurun merges back).
WAIVERS = []
REPORTS = []          # free-form measured facts of a test (e.g. per-stage agreement of an ill-conditioned fixture)


def record_waivers(case, proofs, kind="tie"):
    for stage, frame, gap, bound in proofs:
        WAIVERS.append(dict(case=case, kind=kind, stage=int(stage), frame=int(frame), margin=float(gap), bound=float(bound)))


def record_report(case, **facts):
    REPORTS.append(dict(case=case, **facts))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    import json
    if not WAIVERS and not REPORTS:
        return
    cases = sorted({w["case"] for w in WAIVERS})
    terminalreporter.write_line(f"tie waivers: {len(WAIVERS)} frame(s) in {len(cases)} case(s): " +
                                ", ".join(f"{c} x{sum(1 for w in WAIVERS if w['case'] == c)}" for c in cases))
    for r in REPORTS:      # one compact line per report (long per-stage lists and nested tables stay in the JSON file only)
        short = {k: v for k, v in r.items() if not isinstance(v, (list, dict)) or len(v) <= 4}
        terminalreporter.write_line("report: " + json.dumps(short))
    path = os.environ.get("FC_WAIVER_JSON", os.path.join(ROOT, "gpurun_out", "tie_waivers.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wt") as f:
            json.dump(dict(waivers=WAIVERS, reports=REPORTS), f, indent=1)
    except OSError:
        pass
