#!/usr/bin/env python
"""2-NN filter kernel time by vocabulary size (HIP events around the filter launch, lcd_profile_*): 49k words (headline),
125k (one GPU's shard of config 4) and 1M (config 4 on one GPU), 500 SURF queries."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtabmap_amd  # noqa: E402


def main():
    q = 500
    rng = np.random.default_rng(0)
    out = []
    for n in (49_000, 125_000, 1_000_000):
        v = rng.standard_normal((n, 64)).astype(np.float32)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        qs = v[rng.integers(0, n, q)] + rng.standard_normal((q, 64)).astype(np.float32) * np.float32(0.02)
        eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n, knn_mode=os.environ.get("KNN_MODE") or None)
        ids = np.arange(1, n + 1, dtype=np.int32)
        for a in range(0, n, 250_000):
            eng.vocab_append(v[a:a + 250_000], ids[a:a + 250_000])
        d_q = torch.from_numpy(qs.astype(np.float32)).cuda()
        d_w = torch.zeros(q * 2, dtype=torch.int32, device="cuda")
        d_d = torch.zeros(q * 2, dtype=torch.float32, device="cuda")
        for _ in range(5):
            eng.knn2_dev(d_q.data_ptr(), q, d_w.data_ptr(), d_d.data_ptr())
        eng.synchronize()
        eng.profile_begin(30)
        t0 = time.perf_counter()
        for _ in range(30):
            eng.knn2_dev(d_q.data_ptr(), q, d_w.data_ptr(), d_d.data_ptr())
        eng.synchronize()
        wall = (time.perf_counter() - t0) / 30
        ms, ns, name = eng.profile_read()
        flops = 2.0 * q * n * 64
        out.append({"rows": n, "kernel": name, "filter_ms": ms, "algorithmic_tflops": flops / (ms * 1e-3) / 1e12,
                    "table_gbps": n * 256 / (ms * 1e-3) / 1e9, "knn2_call_ms": wall * 1e3,
                    "fallback_queries": eng.stats()["knn_last_fallback_queries"]})
        eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
