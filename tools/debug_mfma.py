import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtabmap_amd
from rtabmap_amd import synth
N=49000
vocab = synth.vocab_surf(N)
words = synth.zipf_words(200, 500, N, seed=100000)
eng = rtabmap_amd.Engine("f32", 64)
eng.vocab_append(vocab, np.arange(1, N+1, dtype=np.int32))
for i in range(5):
    f = synth.frame_from_signature(vocab, words[i], seed=i)
    ids, d = eng.knn2(f)
    print("frame", i, "fallback queries:", eng.stats()["knn_last_fallback_queries"], "d1 range", d[:,0].min(), d[:,0].max(), "min gap d2-d1", (d[:,1]-d[:,0]).min())
q = synth.queries_surf(vocab, 500)
ids, d = eng.knn2(q)
print("queries_surf fallback:", eng.stats()["knn_last_fallback_queries"])
