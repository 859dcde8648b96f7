#!/usr/bin/env python
"""Per-kernel average of one rocprofv3 --pmc counter from <dir>/*counter_collection.csv (values as rocprofv3 reports them;
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams, see
MI355X_MICROARCH.md section HBM: double it before comparing with a byte count)."""
import collections
import csv
import glob
import sys


def main():
    f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("lcd::", "").replace("void ", "").split("(")[0]
        a = agg[(k, r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    print("# %s" % (sys.argv[2] if len(sys.argv) > 2 else f))
    print("%-60s %-12s %8s %14s" % ("kernel", "counter", "calls", "avg_value"))
    for (k, c), (n, v) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("%-60s %-12s %8d %14.1f" % (k[:60], c, n, v / n))


if __name__ == "__main__":
    main()
