#!/usr/bin/env python
"""Every run of fused launches in a rocprofv3 --kernel-trace CSV (runs split at gaps > 60 us): count, span, and mean / median / p90 / max of the
full launch A's and B's -- which regime of a bench run (timed region, steady leg, per-step-event leg) a change helped or hurt.
    python tools/run_segments.py <kernel_trace.csv> [label]"""
import csv
import sys

import numpy as np


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "frame_a_kernel" in n or "frame_b_kernel" in n:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "A" if "frame_a_kernel" in n else "B", int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))))
    rows.sort()
    runs, cur = [], []
    for r in rows:
        if cur and r[0] - cur[-1][1] > 60000:
            runs.append(cur); cur = []
        cur.append(r)
    if cur:
        runs.append(cur)
    label = sys.argv[2] if len(sys.argv) > 2 else ""
    st = lambda v: "%6.2f %6.2f %6.2f %6.2f" % (np.mean(v), np.median(v), np.percentile(v, 90), np.max(v)) if len(v) else "     -"
    print("%s: run  launches  span_us | A full: n mean median p90 max | B full: n mean median p90 max" % label)
    for i, reg in enumerate(runs):
        if len(reg) < 8:
            continue
        a = [(e - s) / 1000.0 for s, e, k, g in reg if k == "A" and g >= 60000]
        b = [(e - s) / 1000.0 for s, e, k, g in reg if k == "B" and g >= 300000]
        print("  %3d  %5d  %8.1f | %3d %s | %3d %s" % (i, len(reg), (reg[-1][1] - reg[0][0]) / 1000.0, len(a), st(a), len(b), st(b)))


if __name__ == "__main__":
    main()
