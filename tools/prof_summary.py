#!/usr/bin/env python
"""Turns a rocprofv3 (rocpd sqlite) result into the plain-text per-kernel summary kept under profiles/.

usage: python tools/prof_summary.py <results.db> [label]   -> prints name, calls, total us, avg us, % (like --stats)
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    unit = 1.0   # the top_kernels view reports microseconds
    print("# rocprofv3 --kernel-trace --stats summary: %s" % label)
    print("%-90s %10s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in rows:
        short = name.replace("(anonymous namespace)::", "").replace("lcd::", "").replace("void ", "").split("(")[0]
        print("%-90s %10d %14.1f %12.3f %8.2f" % (short[:90], calls, total * unit, avg * unit, pct))


if __name__ == "__main__":
    main()
