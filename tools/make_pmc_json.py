#!/usr/bin/env python
"""profiles/rNN_pmc.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv):
    python tools/make_pmc_json.py <dir_fetch> <dir_write> "<command line profiled>" > profiles/r02_pmc.json
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports KB; FETCH_SIZE counts
64 B per 128-B request on wide coalesced reads, so reads are doubled: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import glob
import json
import sys


def avg(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("lcd::", "").replace("void ", "").split("(")[0].split("<")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def main():
    fe, wr = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only): " + sys.argv[3],
           "unit_note": "rocprofv3 reports KB; gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per 128-B "
                        "request on wide reads, so reads are doubled: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024"}
    for k in sorted(fe, key=lambda k: -fe[k][1] * fe[k][0]):
        if k.startswith("__amd") or k.startswith("at::"):
            continue
        w = wr.get(k, (0, 0.0))[1]
        out[k] = {"calls": fe[k][0], "fetch_kb": round(fe[k][1], 1), "write_kb": round(w, 1),
                  "hbm_bytes_per_launch": int((2 * fe[k][1] + w) * 1024)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
