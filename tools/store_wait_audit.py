#!/usr/bin/env python
"""Audit of a gfx950 assembly listing (hipcc -S --cuda-device-only): where does a wave wait for the acknowledgement of its own stores?
vmcnt is ONE in-order counter for loads, stores and atomics on gfx9: an s_waitcnt vmcnt(n) that is reached with stores (or non-returning
atomics) among the outstanding operations holds the wave for a full memory round trip although no data is awaited.  The walk is linear
(branches ignored), so it over-reports a little; it lists every such wait with the stores in front of it.
    python tools/store_wait_audit.py /tmp/knn_dev.s frame_b_kernel"""
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    in_fn, name = False, ""
    pending = []          # outstanding vm operations in issue order: ("L"|"S", line no, text, source)
    hits = []
    files, loc = {}, ""   # .file / .loc directives (a listing made with -gline-tables-only): source position of every instruction
    for no, ln in enumerate(lines, 1):
        mf = re.match(r'^\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln) or re.match(r'^\s*\.file\s+(\d+)\s+"([^"]+)"', ln)
        if mf:
            files[mf.group(1)] = mf.group(2).split("/")[-1]
            continue
        ml = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if ml:
            loc = "%s:%s" % (files.get(ml.group(1), ml.group(1)), ml.group(2))
            continue
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            in_fn = want in m.group(1)
            name = m.group(1)
            pending = []
            continue
        if not in_fn:
            continue
        t = ln.strip()
        if t.startswith(".Lfunc_end"):
            in_fn = False
            continue
        op = t.split()[0] if t else ""
        if op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")):
            pending.append(("L", no, t, loc))
        elif op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store")):
            pending.append(("S", no, t, loc))
        elif op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
            pending.append(("S" if " sc0" not in t and "glc" not in t else "L", no, t, loc))
        elif op == "s_waitcnt" and "vmcnt" in t:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            done = pending[:max(0, len(pending) - n)]
            stores = [p for p in done if p[0] == "S" and not p[2].startswith("scratch")]
            if stores:
                hits.append((no, t, stores, name, loc))
            pending = pending[max(0, len(pending) - n):] if n else []
    for no, t, stores, name, where in hits:
        print("%s asm %d [%s]: %s  <- %d store(s), first asm %d [%s]: %s" % (name[:40], no, where, t, len(stores), stores[0][1], stores[0][3], stores[0][2][:50]))
    print("%d waits behind stores in functions matching %r" % (len(hits), want))


if __name__ == "__main__":
    main()
