#!/usr/bin/env python
"""Scratch (spill) accesses of one kernel in a gfx950 assembly listing made with -gline-tables-only, counted by the SOURCE FILE the
instruction belongs to (.loc): which part of a fused kernel the compiler parked in scratch memory.
    python tools/scratch_by_file.py /tmp/knn.s frame_b_kernelILb1"""
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    files, loc, on, cnt = {}, "", False, {}
    for ln in open(path):
        mf = re.match(r'^\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln) or re.match(r'^\s*\.file\s+(\d+)\s+"([^"]+)"', ln)
        if mf:
            files[mf.group(1)] = mf.group(2).split("/")[-1]
            continue
        ml = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if ml:
            loc = files.get(ml.group(1), ml.group(1))
            continue
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            on = want in m.group(1)
            continue
        if not on:
            continue
        t = ln.strip()
        if t.startswith(".Lfunc_end"):
            on = False
            continue
        if t.startswith("scratch_"):
            k = (loc, t.split()[0])
            cnt[k] = cnt.get(k, 0) + 1
    for k, v in sorted(cnt.items()):
        print(k, v)
    print("%d scratch accesses in functions matching %r" % (sum(cnt.values()), want))


if __name__ == "__main__":
    main()
