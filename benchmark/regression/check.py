"""Performance-regression gate over all-in-one CSVs.

Reference: python/llm/test/benchmark/csv_to_html.py:37-38, 166, 183-184 (fail the nightly run when first-token or
next-token latency is more than 5 % worse than the stored baseline) and check_results.py:25-66 (every expected
(model, in/out pair, batch) row must be present).

    python benchmark/regression/check.py --baseline profiles/r1_all_in_one_sweep.csv --current new.csv [-t 5.0] [--html out.html]
"""
from __future__ import annotations

import argparse
import csv
import sys
from typing import Dict, List, Tuple

KEY = ("model", "input/output tokens", "batch_size", "low_bit", "test_api")
METRICS = ("1st token avg latency (ms)", "2+ avg latency (ms/token)")


def load(path: str) -> Dict[Tuple, dict]:
    with open(path, newline="") as f:
        return {tuple(r[k] for k in KEY): r for r in csv.DictReader(f)}


def compare(baseline: Dict[Tuple, dict], current: Dict[Tuple, dict], threshold: float = 5.0):
    """-> (rows, missing keys, regressed keys); a row = key + per-metric (baseline, current, percent change)."""
    rows, missing, regressed = [], [], []
    for key, b in baseline.items():
        c = current.get(key)
        if c is None:
            missing.append(key)
            continue
        cells = []
        for m in METRICS:
            bv, cv = float(b[m]), float(c[m])
            pct = (cv - bv) / bv * 100.0 if bv else 0.0
            cells.append((bv, cv, pct))
        rows.append((key, cells))
        if any(pct > threshold for _, _, pct in cells):
            regressed.append(key)
    return rows, missing, regressed


def to_html(rows, threshold: float) -> str:
    out = ["<table border=1><tr>" + "".join(f"<th>{k}</th>" for k in KEY) +
           "".join(f"<th>{m} (baseline)</th><th>{m}</th><th>diff %</th>" for m in METRICS) + "</tr>"]
    for key, cells in rows:
        tds = "".join(f"<td>{k}</td>" for k in key)
        for bv, cv, pct in cells:
            color = "red" if pct > threshold else ("green" if pct < -threshold else "black")
            tds += f"<td>{bv:.3f}</td><td>{cv:.3f}</td><td style='color:{color}'>{pct:+.2f}</td>"
        out.append(f"<tr>{tds}</tr>")
    return "\n".join(out + ["</table>"])


def main(argv: List[str] = None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("-b", "--baseline", required=True)
    ap.add_argument("-c", "--current", required=True)
    ap.add_argument("-t", "--threshold", type=float, default=5.0)
    ap.add_argument("--html")
    a = ap.parse_args(argv)
    rows, missing, regressed = compare(load(a.baseline), load(a.current), a.threshold)
    if a.html:
        open(a.html, "w").write(to_html(rows, a.threshold))
    for k in missing:
        print("MISSING", k)
    for k in regressed:
        print("REGRESSED", k)
    print(f"{len(rows)} rows compared, {len(missing)} missing, {len(regressed)} regressed beyond {a.threshold} %")
    return 1 if missing or regressed else 0


if __name__ == "__main__":
    sys.exit(main())
