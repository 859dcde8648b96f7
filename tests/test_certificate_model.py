"""CPU model of the MFMA-filter certificate (knn_mfma_kernels.hip): the filter sees every distance only up to +-eps and keeps a
handful of keys per partition; the re-rank re-computes the few keys under a threshold exactly and accepts the result only if
`bound - eps > exact second distance`.  The claim: WHENEVER the certificate accepts, the two rows returned are the exact 2-NN
(ties to the lower row).  The model reproduces the data flow of the bf16x3 path -- per (row block, half) partition a top-3 of
truncated keys, per block the best two of the six + the third as bound, threshold tau (1 + 2^-15) + 2 eps -- and lets an
adversary choose the filter's errors inside +-eps (random, or aimed at hiding the true neighbours).  No GPU, no arithmetic of
the kernels: this pins the LOGIC of the proof."""
import numpy as np

INF = np.float64(np.inf)


def truncate_key(score):
    """keys lose the low 7 mantissa bits of the float32 score (towards smaller values for non-negative scores)"""
    b = np.float32(max(score, 0.0)).view(np.uint32)
    return np.uint32(b & np.uint32(0xFFFFFF80)).view(np.float32).astype(np.float64)


def filter_and_rerank(d_exact, score, eps, rows_per_block):
    """returns (ok, best_row, second_row).  d_exact/score: per vocabulary row; +inf score = tombstone."""
    n = len(d_exact)
    kept = []                    # (key, row)
    bound = INF
    for b0 in range(0, n, rows_per_block):
        rows = np.arange(b0, min(b0 + rows_per_block, n))
        six = []
        for half in (0, 1):
            part = rows[((rows >> 2) & 1) == half]                       # the rows one lane sees for this query
            keys = sorted((truncate_key(score[r]), int(r)) for r in part if np.isfinite(score[r]))
            six += keys[:3]                                              # per-lane top-3; what it drops is >= its third key
            if len(keys) > 3:
                pass                                                     # ... which stays in `six` and so reaches the bound below
        six.sort()
        kept += six[:2]
        if len(six) > 2:
            bound = min(bound, six[2][0])                                # third of the block: bounds everything the block dropped
    if len(kept) >= 2:
        tau = sorted(k for k, _ in kept)[1]
    else:
        tau = INF
    thr = tau * (1 + 2.0 ** -15) + 2 * eps if np.isfinite(tau) else INF
    cand = [(d_exact[r], r) for k, r in kept if k <= thr]
    cand.sort()
    best = cand[0] if len(cand) > 0 else None
    second = cand[1] if len(cand) > 1 else None
    ok = True
    if np.isfinite(bound):
        ok = second is not None and bound - eps > second[0]
    return ok, (best[1] if best else -1), (second[1] if second else -1)


def test_accepted_results_are_the_exact_two_nearest():
    rng = np.random.default_rng(0)
    accepted = rejected = 0
    for trial in range(1500):
        n = int(rng.integers(3, 400))
        eps = float(rng.choice([1e-4, 1e-3, 1e-2]))
        kind = rng.integers(0, 4)
        if kind == 0:
            d = rng.uniform(0.05, 2.0, n)
        elif kind == 1:                                                  # a cluster of near ties around the second distance
            d = rng.uniform(0.5, 2.0, n)
            c = rng.integers(0, n, min(n, 8))
            d[c] = 0.3 + rng.uniform(-2 * eps, 2 * eps, len(c))
        elif kind == 2:                                                  # exact ties (duplicates)
            d = rng.uniform(0.2, 2.0, n)
            d[rng.integers(0, n, min(n, 6))] = 0.25
        else:                                                            # neighbours packed in adjacent rows (same block / lane)
            d = rng.uniform(0.5, 2.0, n)
            s0 = int(rng.integers(0, max(1, n - 8)))
            d[s0:s0 + 8] = 0.1 + np.arange(len(d[s0:s0 + 8])) * eps * float(rng.choice([0.1, 1.0, 5.0]))
        d = d.astype(np.float64)
        order = sorted(range(n), key=lambda r: (d[r], r))
        adv = rng.integers(0, 3)
        if adv == 0:
            noise = rng.uniform(-eps, eps, n)
        elif adv == 1:                                                   # hide the true neighbours, promote everybody else
            noise = np.full(n, -eps)
            noise[order[:2]] = eps
        else:                                                            # promote exactly the runner-ups
            noise = np.full(n, eps)
            noise[order[2:6]] = -eps
        score = d + noise
        dead = rng.random(n) < 0.05                                      # tombstones never come back
        dead[order[:1]] = False
        score = np.where(dead, np.inf, score)
        live = [r for r in order if not dead[r]]
        ok, b, s = filter_and_rerank(d, score, eps, rows_per_block=int(rng.choice([32, 64, 192])))
        if ok:
            accepted += 1
            assert b == live[0], (trial, kind, adv)
            assert s == (live[1] if len(live) > 1 else -1), (trial, kind, adv)
        else:
            rejected += 1
    assert accepted > 500 and rejected > 20, (accepted, rejected)         # both outcomes are exercised
