#!/usr/bin/env python
"""Build a variant of liblcd_hip.so with extra compiler flags (timing experiments), next to the product library:
    python tools/build_variant.py scoretiming -DLCD_SCORE_TIMING   ->  rtabmap_amd/liblcd_hip_scoretiming.so
Use it with LCD_LIB_PATH=<that file>.  The product build is untouched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtabmap_amd import build as b  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    objdir = os.path.join(b.HERE, "build", "variant_" + name)
    os.makedirs(objdir, exist_ok=True)
    hipcc = b._hipcc()
    procs, objs = [], []
    for src in b.SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen([hipcc] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj]))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("compile failed")
    out = os.path.join(b.HERE, "liblcd_hip_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)


if __name__ == "__main__":
    main()
