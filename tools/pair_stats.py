#!/usr/bin/env python
"""What the driver's timed region is made of, from a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W --no-cpu-baseline --no-extras --no-pmc`:
    python tools/pair_stats.py <kernel_trace.csv> [label] [--dump]      (--dump: also every launch of the region: start, duration, gap in front, grid)
The fused launches (frame_a_kernel*, frame_b_kernel*) are split into runs at gaps > 60 us; the TIMED region is the first run of at least
2 K + 4 launches (the warm-up's run is shorter, the per-step-event leg behind it has a gap in front of every launch).  Printed (us):
  span      first launch start -> last launch end of the timed region (the wall clock adds the enqueue latency in front of the first launch)
  lead      gap in front of the region's first and second launch (an idle queue: host enqueue + launch latency)
  gaps      sum of the other gaps inside the region (profiling events of the first steps)
  A / B     mean and median duration of the FULL launches (a launch A with a filter, a launch B with re-rank + scoring workgroups), the B's also
            split at 17 us: frames whose predecessor created words (growth) against revisits
  drain     the launches of the last three pairs"""
import csv
import statistics
import sys


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
    dump = "--dump" in sys.argv
    if dump:
        sys.argv.remove("--dump")
    label = sys.argv[2] if len(sys.argv) > 2 else ""
    big = [x for x in runs if len(x) >= 40]
    if not big:
        print(label, "no timed region found; runs:", [len(x) for x in runs]); return
    reg = big[0]
    us = lambda ns: ns / 1000.0
    span = us(reg[-1][1] - reg[0][0])
    gaps = [us(reg[i][0] - reg[i - 1][1]) for i in range(1, len(reg))]
    lead = gaps[0] if gaps else 0.0
    a_full = [us(e - s) for s, e, k, g in reg if k == "A" and g >= 60000]
    b_full = [us(e - s) for s, e, k, g in reg if k == "B" and g >= 300000]
    b_grow = [d for d in b_full if d >= 17.0]
    b_rev = [d for d in b_full if d < 17.0]
    f = lambda v: "%.2f/%.2f (n=%d)" % (statistics.mean(v), statistics.median(v), len(v)) if v else "-"
    print("%-26s span %7.1f  lead %5.1f  other-gaps %5.1f  launches %d | A %s | B %s  growth %s  revisit %s | drain %s" % (
        label, span, lead, sum(gaps[1:]), len(reg), f(a_full), f(b_full), f(b_grow), f(b_rev),
        " ".join("%s%.1f" % (k, us(e - s)) for s, e, k, g in reg[-6:])))
    if dump:
        print("# the launches of the timed region: start (us after the first), duration, gap in front, kind, grid (threads)")
        for i, (s, e, k, g) in enumerate(reg):
            print("%9.2f  dur %6.2f  gap %6.2f  %s  grid %7d%s" % (us(s - reg[0][0]), us(e - s), gaps[i - 1] if i else 0.0, k, g,
                                                              "   <- drain" if i >= len(reg) - 6 else ""))


if __name__ == "__main__":
    main()
