"""CPU emulation of the f32 MFMA filter's arithmetic (knn_mfma_filter_kernel: one fp32 FMA chain over |v|^2, |q|^2 and the
64 products v_k * (-2 q_k), k-steps pairing element t with element 32 + t) against the reference's squared L2 in rtflann's
order (dist.h:150-177), to check eps_for() = (3.5 D + 16) * 2^-24 * 1.25 * (|q|^2 + max |v|^2) on random, wide-range and
cancellation-heavy inputs.  fma(a, b, c) is emulated as float32(float64(a) * float64(b) + float64(c)) (the product is exact in
float64; the rare double rounding is far below the bound being tested)."""
import numpy as np

U = 2.0 ** -24


def fma32(a, b, c):
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def filter_score(v, q):
    vn = np.float32(0.0)
    for x in v:                                  # row_norm_kernel: fmaf chain
        vn = fma32(x, x, vn)
    qn0 = np.float32(0.0)
    qn1 = np.float32(0.0)
    for k in range(32):                          # the kernel sums each half separately and adds the two
        qn0 = fma32(q[k], q[k], qn0)
        qn1 = fma32(q[32 + k], q[32 + k], qn1)
    qn = np.float32(qn0 + qn1)
    acc = fma32(vn, np.float32(1.0), np.float32(0.0))
    acc = fma32(np.float32(1.0), qn, acc)
    b = (np.float32(-2.0) * q).astype(np.float32)
    for t in range(32):
        acc = fma32(v[t], b[t], acc)
        acc = fma32(v[32 + t], b[32 + t], acc)
    return float(acc), float(qn), float(vn)


def ref_l2(v, q):
    res = np.float32(0.0)
    for g in range(0, 64, 4):
        d = (v[g:g + 4] - q[g:g + 4]).astype(np.float32)
        t = np.float32(d[0] * d[0])
        t = np.float32(t + np.float32(d[1] * d[1]))
        t = np.float32(t + np.float32(d[2] * d[2]))
        t = np.float32(t + np.float32(d[3] * d[3]))
        res = np.float32(res + t)
    return float(res)


def test_eps_for_covers_the_f32_filter():
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(600):
        kind = trial % 4
        if kind == 0:
            v = rng.standard_normal(64); q = rng.standard_normal(64)
        elif kind == 1:                          # near-identical (cancellation: the distance is tiny next to the norms)
            v = rng.standard_normal(64); q = v * (1 + 1e-4 * rng.standard_normal(64))
        elif kind == 2:                          # wide dynamic range
            v = rng.standard_normal(64) * np.exp(rng.uniform(-6, 6, 64)); q = v + rng.standard_normal(64) * np.exp(rng.uniform(-6, 6, 64))
        else:                                    # unit-norm SURF-like
            v = np.abs(rng.standard_normal(64)); v /= np.linalg.norm(v); q = np.abs(rng.standard_normal(64)); q /= np.linalg.norm(q)
        v = v.astype(np.float32); q = q.astype(np.float32)
        s, qn, vn = filter_score(v, q)
        eps = (3.5 * 64 + 16) * U * 1.25 * (qn + vn)
        err = abs(s - ref_l2(v, q))
        worst = max(worst, err / eps)
        assert err <= eps, (trial, kind, err, eps)
    assert 0.0 < worst < 0.5, worst             # room to spare, as the GPU measurement (knn_max_err_ratio 0.21) says
