"""CPU check of the split-error term of eps_bf16() (rtabmap_amd/csrc/knn_mfma_kernels.hip): with x = hi + lo + d, hi = bf16(x),
lo = bf16(x - hi), the bf16x3 filter replaces q.v by qh.vh + qh.vl + ql.vh.  The certificate charges
3.1 * 2^-16 * (|q|^2 + |v|^2) for what that neglects (on the score -2 q.v).  Emulated here in numpy (bf16 = float32 rounded to
nearest-even at bit 16, products and sums in float64 so that only the split error is measured) on random, wide-range and
adversarial inputs."""
import numpy as np


def bf16_rne(x):
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16_rne(x)
    lo = bf16_rne((x.astype(np.float32) - hi).astype(np.float32))
    return hi.astype(np.float64), lo.astype(np.float64)


def worst_ratio(q, v):
    """max over pairs of |(-2 q.v) - (-2 (qh.vh + qh.vl + ql.vh))| / (|q|^2 + |v|^2)"""
    qh, ql = split(q)
    vh, vl = split(v)
    q64, v64 = q.astype(np.float64), v.astype(np.float64)
    exact = q64 @ v64.T
    approx = qh @ vh.T + qh @ vl.T + ql @ vh.T
    denom = (q64 ** 2).sum(1)[:, None] + (v64 ** 2).sum(1)[None, :]
    return float((2.0 * np.abs(exact - approx) / denom).max())


def test_split_error_stays_inside_the_charged_bound():
    rng = np.random.default_rng(0)
    bound = 3.1 * 2.0 ** -16
    cases = []
    a = rng.standard_normal((300, 64)).astype(np.float32)
    cases.append((a / np.linalg.norm(a, axis=1, keepdims=True), a[::-1] / np.linalg.norm(a[::-1], axis=1, keepdims=True)))
    w = (rng.standard_normal((300, 64)) * np.exp(rng.uniform(-8, 8, (300, 64)))).astype(np.float32)
    cases.append((w, w[rng.permutation(300)]))
    cases.append((w, (w * (1 + 1e-3 * rng.standard_normal(w.shape))).astype(np.float32)))      # near-identical pairs
    # adversarial: every component sits just below a bf16 rounding boundary (largest |lo|), signs aligned so that the errors add
    m = (1.0 + (2.0 ** -8) * (1 - 2.0 ** -9)) * 2.0 ** rng.integers(-3, 3, (200, 64))
    adv = m.astype(np.float32)
    cases.append((adv, adv[::-1].copy()))
    cases.append((adv, adv.copy()))
    worst = max(worst_ratio(q, v) for q, v in cases)
    assert 0.0 < worst < bound, (worst, bound)
    # the bound is not vacuous: the adversarial case comes within an order of magnitude of it
    assert worst > bound / 20, (worst, bound)


def test_bf16_rne_helper():
    x = np.array([1.0, 1.00390625, 1.005859375, -2.5, 3.0e38, 1e-40], np.float32)
    hi = bf16_rne(x)
    assert hi[0] == 1.0 and hi[3] == -2.5
    assert hi[1] in (np.float32(1.0), np.float32(1.0078125))      # tie: to even
    assert abs(float(hi[2]) - float(x[2])) <= 2.0 ** -8
    hi2, lo2 = split(x[:4])
    assert np.all(np.abs(x[:4].astype(np.float64) - hi2 - lo2) <= 2.0 ** -17 * np.abs(x[:4]))
