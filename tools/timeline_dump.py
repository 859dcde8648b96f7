#!/usr/bin/env python
"""Kernel timeline of a rocprofv3 --kernel-trace CSV: python tools/timeline_dump.py <csv> [first | kernel name] [count]
start offset, duration, gap to the previous kernel (us), grid size, name -- every kernel, in start order."""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?")))
    rows.sort()
    if len(sys.argv) > 2 and not sys.argv[2].lstrip("-").isdigit():      # a kernel name: start at its first launch (e.g. frame_a_kernel)
        name = sys.argv[2]
        hits = [i for i, r in enumerate(rows) if name in r[2]]
        sys.argv[2] = str(hits[0] if hits else 0)
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    if first < 0:
        first = max(0, len(rows) + first)
    t0 = rows[first][0]
    prev_end = None
    print("# %d kernels in the trace; showing %d from #%d" % (len(rows), min(count, len(rows) - first), first))
    for s, e, name, g, w in rows[first:first + count]:
        short = name.replace("(anonymous namespace)::", "").replace("lcd::", "").replace("void ", "").split("(")[0]
        gap = (s - prev_end) / 1000.0 if prev_end is not None else 0.0
        print("%9.2f  dur %7.2f  gap %7.2f  grid %8s x %4s  %s" % ((s - t0) / 1000.0, (e - s) / 1000.0, gap, g, w, short[:60]))
        prev_end = e


if __name__ == "__main__":
    main()
