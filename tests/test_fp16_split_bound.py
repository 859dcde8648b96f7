"""CPU check of the error terms an fp16 matrix-core filter would charge (tools/fp16_filter_model.py; DESIGN.md section 7a: the measured
alternative to the bf16x3 filter that north_star names).  With x16 = fp16(x) (round to nearest even, u = 2^-11) the score
|q|^2 + |v|^2 - 2 q.v computed from rounded operands differs from the exact one by at most
    one product    q16.v16                   (2u + u^2) (|q|^2 + |v|^2)
    two products   q16.(v16 + vlo16)         (u + 2u^2) (|q|^2 + |v|^2)          (the query's rounding remains)
    three products hi.hi + hi.lo + lo.hi     3.1 u^2    (|q|^2 + |v|^2)
for operands inside fp16's normal range (|x| in [2^-14, 65504]); emulated in numpy with products and sums in float64 so that only the
operand rounding is measured, on random, wide-range and adversarial inputs."""
import numpy as np


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float64)


def worst_ratios(q, v):
    q64, v64 = q.astype(np.float64), v.astype(np.float64)
    qh, vh = f16(q), f16(v)
    ql, vl = f16(q64 - qh), f16(v64 - vh)
    exact = q64 @ v64.T
    denom = (q64 ** 2).sum(1)[:, None] + (v64 ** 2).sum(1)[None, :]
    r = lambda approx: float((2.0 * np.abs(exact - approx) / denom).max())   # noqa: E731
    return r(qh @ vh.T), r(qh @ vh.T + qh @ vl.T), r(qh @ vh.T + qh @ vl.T + ql @ vh.T)


def test_fp16_operand_rounding_stays_inside_the_charged_bounds():
    rng = np.random.default_rng(0)
    u = 2.0 ** -11
    bounds = (2 * u + u * u, u + 2 * u * u, 3.1 * u * u)
    cases = []
    a = rng.standard_normal((300, 64)).astype(np.float32)
    cases.append((a / np.linalg.norm(a, axis=1, keepdims=True), a[::-1] / np.linalg.norm(a[::-1], axis=1, keepdims=True)))
    w = (rng.standard_normal((300, 64)) * np.exp(rng.uniform(-4, 4, (300, 64)))).astype(np.float32)      # fp16's normal range
    w = np.sign(w) * np.clip(np.abs(w), 2.0 ** -13, 1000.0)
    cases.append((w, w[rng.permutation(300)]))
    cases.append((w, (w * (1 + 1e-3 * rng.standard_normal(w.shape))).astype(np.float32)))                # near-identical pairs
    # adversarial: every component just below an fp16 rounding boundary (largest relative error), signs aligned so that the errors add
    m = ((1.0 + (2.0 ** -11) * (1 - 2.0 ** -9)) * 2.0 ** rng.integers(-3, 3, (200, 64))).astype(np.float32)
    cases.append((m, m[::-1].copy()))
    cases.append((m, m.copy()))
    worst = np.array([worst_ratios(q, v) for q, v in cases]).max(axis=0)
    for w_, b in zip(worst, bounds):
        assert 0.0 < w_ < b, (worst, bounds)
    assert worst[0] > bounds[0] / 20 and worst[2] > bounds[2] / 40, (worst, bounds)        # the bounds are not vacuous


def test_fp16_filter_on_the_bench_data_needs_few_candidates_and_no_redo():
    """The consequence for the exact 2-NN on the headline data (unit-norm 64-float words, frames that revisit a place): with the
    one-product score the re-rank re-computes ~2.3 rows per query instead of ~2.0, and no query loses its certificate -- the numbers
    DESIGN.md quotes next to the kernel times of the one-product build."""
    from rtabmap_amd import synth
    n_words, n_q, strip = 20000, 200, 7 * 32
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(8, n_q, n_words, seed=100000)
    q = synth.frame_from_signature(vocab, words[1], seed=1001)
    u = 2.0 ** -11
    q64, v64 = q.astype(np.float64), vocab.astype(np.float64)
    qn, vn = (q64 ** 2).sum(1)[:, None], (v64 ** 2).sum(1)[None, :]
    d2 = qn + vn - 2.0 * (q64 @ v64.T)
    score = qn + vn - 2.0 * (f16(q) @ f16(vocab).T)
    eps = (2 * u + u * u) * (qn + vn).max(axis=1)
    assert np.abs(score - d2).max() < 0.5 * eps.min()                  # (the re-rank distrusts a filter that uses half its budget)
    n_strips = (n_words + strip - 1) // strip
    s = np.pad(score, ((0, 0), (0, n_strips * strip - n_words)), constant_values=np.inf).reshape(n_q, n_strips, strip)
    part = np.sort(np.partition(s, 2, axis=2)[:, :, :3], axis=2)
    kept, bound = part[:, :, :2].reshape(n_q, -1), part[:, :, 2].min(axis=1)
    tau = np.sort(kept, axis=1)[:, 1]
    cand = (kept <= (tau * (1 + 2.0 ** -15) + 2 * eps)[:, None]).sum(axis=1)
    second = np.sort(d2, axis=1)[:, 1]
    assert cand.mean() < 4.0 and cand.max() <= 16
    assert (bound - eps > second).mean() > 0.97
