#!/usr/bin/env python
"""Durations and gaps of the two fused launches of a pipelined frame from a rocprofv3 --kernel-trace CSV:
    python tools/dispatch_gaps.py <kernel_trace.csv>
prints the median / p90 duration of frame_a_kernel and frame_b_kernel, the median gap end(A) -> start(B) and end(B) -> start(next A)
and the median period of a frame (start(A) -> start(next A)) over the steady part of the run (launches that carry every stage)."""
import csv
import sys

import numpy as np


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    seq = [(s, e, "A" if "frame_a_kernel" in n else "B") for s, e, n in rows if "frame_a_kernel" in n or "frame_b_kernel" in n]
    dur = {"A": [], "B": []}
    gap_ab, gap_ba, period = [], [], []
    for i in range(1, len(seq) - 1):
        s, e, k = seq[i]
        ps, pe, pk = seq[i - 1]
        ns, ne, nk = seq[i + 1]
        if k == "A" and pk == "B" and nk == "B":
            g = (s - pe) / 1000.0
            if g < 20.0:                                   # back-to-back launches only (not the pauses between the run's phases)
                gap_ba.append(g)
                dur["A"].append((e - s) / 1000.0)
        if k == "B" and pk == "A" and nk == "A":
            g = (s - pe) / 1000.0
            if g < 20.0:
                gap_ab.append(g)
                dur["B"].append((e - s) / 1000.0)
                if (ns - ps) / 1000.0 < 100.0:
                    period.append((ns - ps) / 1000.0)
    q = lambda v, p: float(np.percentile(np.array(v), p)) if v else float("nan")
    print("# rocprofv3 --kernel-trace time stamps of a pipelined stream (us); %d frames" % len(period))
    for k in "AB":
        print("frame_%s_kernel duration   median %6.2f  p10 %6.2f  p90 %6.2f" % (k.lower(), q(dur[k], 50), q(dur[k], 10), q(dur[k], 90)))
    print("gap end(A) -> start(B)     median %6.2f  p10 %6.2f  p90 %6.2f" % (q(gap_ab, 50), q(gap_ab, 10), q(gap_ab, 90)))
    print("gap end(B) -> start(A)     median %6.2f  p10 %6.2f  p90 %6.2f" % (q(gap_ba, 50), q(gap_ba, 10), q(gap_ba, 90)))
    print("frame period (A -> next A) median %6.2f  p10 %6.2f  p90 %6.2f" % (q(period, 50), q(period, 10), q(period, 90)))


if __name__ == "__main__":
    main()
