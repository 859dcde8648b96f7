#!/usr/bin/env python
"""Per-frame kernel timeline from a rocprofv3 --kernel-trace CSV: start offset, duration and gap to the previous kernel (us)."""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # last complete frame: from the last filter kernel backwards one frame
    idx = [i for i, r in enumerate(rows) if "filter_kernel" in r[2]]
    n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    a, b = idx[-(n_show + 1)], idx[-1]
    t0 = rows[a][0]
    prev_end = None
    for s, e, name in rows[a:b]:
        short = name.replace("(anonymous namespace)::", "").replace("lcd::", "").replace("void ", "").split("(")[0]
        gap = (s - prev_end) / 1000.0 if prev_end is not None else 0.0
        print("%9.2f  dur %7.2f  gap %6.2f  %s" % ((s - t0) / 1000.0, (e - s) / 1000.0, gap, short[:70]))
        prev_end = e


if __name__ == "__main__":
    main()
