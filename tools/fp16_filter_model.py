#!/usr/bin/env python
"""What would an fp16 matrix-core filter cost the exact 2-NN?  (north_star: "MFMA fp16 only for the SURF L2-distance GEMM".)

The filter only RANKS: per strip of vocabulary rows it keeps the two best scores of a query and the third as the strip's bound; the
re-rank re-computes exactly every kept key whose score is <= tau (1 + 2^-15) + 2 eps (tau = second-smallest kept score) and certifies
the result when (smallest bound) - eps > exact second distance, else the query is redone by the exact scan.  eps must bound |score -
distance|.  With the operands rounded to fp16 (u = 2^-11):
    one product   q16 . v16                      |error on -2 q.v| <= 2 (2u + u^2) |q||v|  <= (2u + u^2)(|q|^2 + |v|^2)
    two products  q16 . (v16 + vlo16)            the query's rounding remains: <= 2 (u + ..)|q||v|
    three products (hi.hi + hi.lo + lo.hi), fp16 <= ~2 * 3 u^2 |q||v|   (the bf16x3 filter in the product: 3.1 * 2^-16 (|q|^2 + |v|^2))
This script measures, on the bench's data (49k unit-norm SURF-like words, 500 queries that revisit a place), per variant: the eps the
model charges, the largest error actually seen, the candidates per query the re-rank would have to re-compute, and the share of
queries whose certificate fails (each one costs an exact scan of the whole vocabulary).  numpy only (fp16 = IEEE half, products and
sums in float64 so that only the operand rounding is measured); the kernel time of a one- / two-product filter is measured on the GPU
with -DLCD_MFMA_ABLATE=4 / 5 builds (tools/bench_knn_sizes.py): same MFMA rate for fp16 and bf16 on gfx950."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtabmap_amd import synth  # noqa: E402


def bf16(x):
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def variants(q, v):
    """name -> (approximate q.v [Q, N] float64, eps factor on (|q|^2 + |v|^2))"""
    q64, v64 = q.astype(np.float64), v.astype(np.float64)
    u = 2.0 ** -11
    qh, vh = f16(q), f16(v)
    ql, vl = f16(q - qh), f16(v - vh)
    bqh, bvh = bf16(q), bf16(v)
    bql, bvl = bf16(q - bqh), bf16(v - bvh)
    d = lambda a, b: a.astype(np.float64) @ b.astype(np.float64).T   # noqa: E731
    out = {
        "fp16 x1 (qh.vh)": (d(qh, vh), 2 * u + u * u),
        "fp16 x2 (qh.vh + qh.vl)": (d(qh, vh) + d(qh, vl), u + 2 * u * u),
        "fp16 x3 (+ ql.vh)": (d(qh, vh) + d(qh, vl) + d(ql, vh), 3.1 * u * u),
        "bf16 x3 (the product's filter)": (d(bqh, bvh) + d(bqh, bvl) + d(bql, bvh), 3.1 * 2.0 ** -16),
    }
    return out, q64 @ v64.T


def main():
    n_words, n_q, strip = 49000, 500, 7 * 32
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(64, n_q, n_words, seed=100000)
    rows = []
    for f in range(4):
        q = synth.frame_from_signature(vocab, words[f], seed=1000 + f)
        app, exact = variants(q, vocab)
        qn = (q.astype(np.float64) ** 2).sum(1)[:, None]
        vn = (vocab.astype(np.float64) ** 2).sum(1)[None, :]
        d2 = qn + vn - 2.0 * exact
        d2_sorted = np.sort(d2, axis=1)
        second = d2_sorted[:, 1]
        for name, (dot, fac) in app.items():
            score = qn + vn - 2.0 * dot
            eps = fac * (qn + vn).max(axis=1)                                   # (the kernel charges the largest |v|^2 of the vocabulary)
            err = np.abs(score - d2).max(axis=1)
            n_strips = (n_words + strip - 1) // strip
            pad = n_strips * strip - n_words
            s = np.pad(score, ((0, 0), (0, pad)), constant_values=np.inf).reshape(n_q, n_strips, strip)
            part = np.partition(s, 2, axis=2)[:, :, :3]
            part.sort(axis=2)
            kept = part[:, :, :2].reshape(n_q, -1)
            bound = part[:, :, 2].min(axis=1)
            tau = np.sort(kept, axis=1)[:, 1]
            thr = tau * (1 + 2.0 ** -15) + 2 * eps
            cand = (kept <= thr[:, None]).sum(axis=1)
            fail = ~(bound - eps > second)
            rows.append((name, float(eps.mean()), float(err.max()), float(np.mean(err / eps)), float(cand.mean()), int(cand.max()), float(fail.mean())))
    out = {}
    for name in dict.fromkeys(r[0] for r in rows):
        r = [x for x in rows if x[0] == name]
        out[name] = {"eps_charged": np.mean([x[1] for x in r]), "max_error_seen": max(x[2] for x in r), "mean_error_over_eps": np.mean([x[3] for x in r]),
                     "candidates_per_query_mean": np.mean([x[4] for x in r]), "candidates_per_query_max": max(x[5] for x in r),
                     "certificate_failures": np.mean([x[6] for x in r])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
