#!/usr/bin/env python
"""Instruction mix of the loops of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only [-gline-tables-only]).

A loop is the span between a label and the LAST backward branch to it; for every loop that contains matrix-core instructions (or, with
--all, every loop of at least --min instructions) the instructions are counted by class: MFMA, VALU, SALU, LDS, vector memory, waits,
barriers.  The VALU budget of a loop is what bounds a wave that already hides its matrix-core work: one VALU instruction is four cycles
of a wave64 on a SIMD, whatever the matrix pipe does meanwhile (tools/ubench/mfma_valu_grain.hip measures the overlap).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -S --cuda-device-only \\
          rtabmap_amd/csrc/knn_mfma_kernels.hip -o /tmp/knn.s
    python tools/isa_loop_histogram.py /tmp/knn.s 'frame_a_kernelILi1'
"""
import argparse
import collections
import re


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(("ds_", "buffer_load_dword_lds")):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_body(lines, want):
    out, on = [], False
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            on = want in m.group(1)
            continue
        if not on:
            continue
        t = ln.strip()
        if t.startswith(".Lfunc_end"):
            break
        if not t or t.startswith((";", ".loc", ".cfi", ".file", ".Ltmp", ".p2align", ".section")):
            continue
        out.append(t)
    return out


def loops(body):
    """(label, first index, last index) of every label that some later branch jumps back to."""
    at = {}
    for i, t in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            at[m.group(1)] = i
    spans = {}
    for i, t in enumerate(body):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", t)
        if m and m.group(1) in at and at[m.group(1)] <= i:
            spans[m.group(1)] = (at[m.group(1)], i)
    return sorted((a, b, lbl) for lbl, (a, b) in spans.items())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listing")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--all", action="store_true", help="every loop, not only those with matrix-core instructions")
    ap.add_argument("--min", type=int, default=32)
    ap.add_argument("--ops", action="store_true", help="also the opcode histogram of each reported loop")
    a = ap.parse_args()
    body = kernel_body(open(a.listing).read().splitlines(), a.kernel)
    if not body:
        raise SystemExit("no kernel matching %r in %s" % (a.kernel, a.listing))
    print("%s: %d instructions and labels" % (a.kernel, len(body)))
    for first, last, lbl in loops(body):
        ins = [t.split()[0] for t in body[first:last + 1] if not t.endswith(":") and not re.match(r"^\.LBB\w+:", t)]
        cls = collections.Counter(classify(op) for op in ins)
        if not a.all and not cls["mfma"]:
            continue
        if len(ins) < a.min:
            continue
        print("loop %s [%d..%d]: %d instructions | mfma %d valu %d salu %d lds %d vmem %d wait %d barrier %d branch %d | VALU issue >= %d cycles per trip"
              % (lbl, first, last, len(ins), cls["mfma"], cls["valu"], cls["salu"], cls["lds"], cls["vmem"], cls["wait"], cls["barrier"],
                 cls["branch"], 4 * cls["valu"]))
        if a.ops:
            for op, n in collections.Counter(ins).most_common():
                print("    %5d %s" % (n, op))


if __name__ == "__main__":
    main()
