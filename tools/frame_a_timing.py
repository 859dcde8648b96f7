#!/usr/bin/env python
"""Timing experiment: where launch A of a pipelined frame spends its time.  Build a variant with both stamp sets,
    python tools/build_variant.py atiming -DLCD_MFMA_TIMING -DLCD_TAIL_TIMING
and run with LCD_LIB_PATH=rtabmap_amd/liblcd_hip_atiming.so.  Prints (us after the first stamp of the launch) when the filter waves
entered / left the kernel and when the decision-loop and registration workgroups started / finished, for the last fused launch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtabmap_amd  # noqa: E402
from rtabmap_amd import capi, synth  # noqa: E402


def main():
    n_words, q, n_sig = int(os.environ.get("N_WORDS", "49000")), 500, int(os.environ.get("N_SIG", "100000"))
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(n_sig, q, n_words, seed=100000)
    eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n_words + 65536, sig_capacity=n_sig + 4096, pipeline=True)
    for kv in filter(None, os.environ.get("LCD_BENCH_OPTS", "").split(",")):
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    opts = dict(kv.split("=") for kv in filter(None, os.environ.get("LCD_BENCH_OPTS", "").split(",")))
    strip = int(opts.get("strip_tiles", 0))
    eng.vocab_append(vocab, np.arange(1, n_words + 1, dtype=np.int32))
    eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), words.reshape(-1))
    d_words = torch.zeros(q, dtype=torch.int32, device="cuda")
    cap = n_sig + 4096
    d_like = torch.zeros(cap, dtype=torch.float32, device="cuda")
    lib = capi.load()
    # N_FRAMES distinct frames (8: after the first pass every frame is a revisit that creates no word; 128 with APPEND=1: every frame of
    # the run creates ~150 words, the growth phase of an incremental dictionary)
    n_distinct = int(os.environ.get("N_FRAMES", "8"))
    frames = [torch.from_numpy(synth.frame_from_signature(vocab, words[i * 11], seed=i)).cuda() for i in range(n_distinct)]
    res = []
    gi = 0
    for rep in range(6):
        n = 10 + rep
        for i in range(n):
            gi += 1
            eng.frame_dev(frames[(gi if n_distinct > 8 else i) % n_distinct].data_ptr(), q, n_sig + 1000 * rep + 1 + i, float(n_sig + 1), d_words.data_ptr(), d_like.data_ptr(), cap,
                          incremental=True, new_words_compared=True, nndr=0.8, first_new_word_id=n_words + 1 + (100 * rep + i) * q,
                          append_new_words=bool(os.environ.get("APPEND")))
            eng.sig_remove(1 + 20 * rep + i)
        torch.cuda.synchronize()                       # NOT eng.synchronize(): the stamps of the last fused launch must survive
        tb = (ctypes.c_ulonglong * 64)()
        assert lib.lcd_debug_tail_timing_pipe(tb) == 0
        mb = (ctypes.c_ulonglong * (4 * 4096))()
        lib.lcd_debug_mfma_timing.restype = ctypes.c_int
        assert lib.lcd_debug_mfma_timing(mb, 4 * 4096) == 0
        t = np.frombuffer(mb, dtype=np.uint64).reshape(-1, 4).astype(np.float64)
        wg_of = np.arange(t.shape[0]) // 4
        keep = (t[:, 0] > 0) & (t[:, 3] >= t[:, 0])
        keep &= t[:, 3] > t[keep, 3].max() - 20000
        late = keep & (t[:, 0] > t[keep, 0].min() + 500)
        if rep == 5:
            print("workgroups entering > 5 us after the first:", sorted(set(wg_of[late].tolist())))
        t = t[keep]
        # keep the waves of the last launch only (stamps within 200 us of the newest one)
        newest = t[:, 3].max()
        t = t[t[:, 3] > newest - 20000]
        tail = np.array(tb[:8], dtype=np.float64)
        t0 = min(t[:, 0].min(), tail[0] if tail[0] > newest - 20000 else 1e30, tail[4] if tail[4] > newest - 20000 else 1e30)
        f = (t - t0) / 100.0
        tl = (tail - t0) / 100.0
        res.append((f, tl))
        if hasattr(lib, "lcd_debug_a_timing") and rep == 5:
            ab = (ctypes.c_ulonglong * (2 * 4096))()
            assert lib.lcd_debug_a_timing(ab, 2 * 4096) == 0
            b = np.frombuffer(ab, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
            idx = np.nonzero(b[:, 1] > b[:, 1].max() - 20000)[0]
            b = (b[idx] - b[idx, 0].min()) / 100.0
            n_t = (n_words + 31) // 32                                     # the engine's plan (build_knn): the distance tiles keep their compute units
            w = -(-n_t // (256 - 36)) if not strip else strip
            n_f = -(-n_t // w) if w <= 8 else None
            print("launch A: %d workgroups stamped" % len(idx))
            groups = [("decision loop", idx == 0), ("registration", idx == 1)]
            if n_f:
                a0 = 2 + n_f
                T = (q + 63) // 64
                n_tl = T * (T + 1) // 2 + (T * T if int(opts.get("cross_frame_tiles", 0)) else 0)   # same-frame tiles (+ cross-frame ones)
                n_qs = max(1, min((T * 64 * 8 + 255) // 256, 32))                                   # launch_frame_a: one item per thread
                # round 6: the shadow-score workgroups (one per 32-row tile of the frame before, per block of 512 queries) sit between the tiles and the pre-split
                n_sh = (T * 2) * ((q + 511) // 512) if (os.environ.get("APPEND") and int(opts.get("shadow_rows", 1)) and not int(opts.get("cross_frame_tiles", 0))) else 0
                a1 = a0 + n_tl + n_sh
                groups += [("filter", (idx >= 2) & (idx < a0)), ("distance tiles", (idx >= a0) & (idx < a0 + n_tl)), ("shadow scores", (idx >= a0 + n_tl) & (idx < a1)),
                           ("query pre-split", (idx >= a1) & (idx < a1 + n_qs)), ("redo helpers", idx >= a1 + n_qs)]
            for nme, m in groups:
                if m.sum():
                    st, en = b[m, 0], b[m, 1]
                    print("  %-14s n %4d start min %5.2f median %5.2f max %5.2f | end median %5.2f p90 %5.2f max %5.2f | duration median %5.2f max %5.2f" %
                          (nme, m.sum(), st.min(), np.median(st), st.max(), np.median(en), np.percentile(en, 90), en.max(), np.median(en - st), (en - st).max()))
        if hasattr(lib, "lcd_debug_b_timing") and rep == 5:
            bb = (ctypes.c_ulonglong * (2 * 4096))()
            assert lib.lcd_debug_b_timing(bb, 2 * 4096) == 0
            b = np.frombuffer(bb, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
            idx = np.nonzero(b[:, 1] > b[:, 1].max() - 20000)[0]
            b = (b[idx] - b[idx, 0].min()) / 100.0
            nr = ((q + 1) // 2 + 7) & ~7
            print("launch B: %d workgroups stamped (%d re-rank, %d scoring)" % (len(idx), (idx < nr).sum(), (idx >= nr).sum()))
            for nme, m in (("re-rank", idx < nr), ("scoring", idx >= nr)):
                if m.sum():
                    st, en = b[m, 0], b[m, 1]
                    print("  %-8s start min %5.2f median %5.2f max %5.2f | end median %5.2f p90 %5.2f max %5.2f | duration median %5.2f p90 %5.2f max %5.2f" %
                          (nme, st.min(), np.median(st), st.max(), np.median(en), np.percentile(en, 90), en.max(), np.median(en - st),
                           np.percentile(en - st, 90), (en - st).max()))
        if hasattr(lib, "lcd_debug_rr_timing") and rep == 5:
            rr = (ctypes.c_ulonglong * (8 * 512))()
            assert lib.lcd_debug_rr_timing(rr, 8 * 512) == 0
            t8raw = np.frombuffer(rr, dtype=np.uint64).reshape(-1, 8).astype(np.float64)[:250]
            if hasattr(lib, "lcd_debug_b_timing"):
                bb3 = (ctypes.c_ulonglong * (2 * 4096))()
                assert lib.lcd_debug_b_timing(bb3, 2 * 4096) == 0
                b3 = np.frombuffer(bb3, dtype=np.uint64).reshape(-1, 2).astype(np.float64)[:250]
                if os.environ.get("RR_SUBSTAMP"):
                    print("re-rank sub-stamps after the barrier: first reduction %.2f, second reduction %.2f, results written %.2f us" %
                          tuple(np.median(t8raw[:, i] - t8raw[:, 6]) / 100.0 for i in (4, 5, 7)))
                print("re-rank workgroups: first phase stamp - workgroup start median %.2f us; results written - barrier %.2f; workgroup end - barrier stamp median %.2f us" %
                      (np.median(t8raw[:, 0] - b3[:, 0]) / 100.0, np.median(t8raw[:, 7] - t8raw[:, 6]) / 100.0, np.median(b3[:, 1] - t8raw[:, 6]) / 100.0))
            t8 = (t8raw[:, :7] - t8raw[:, :1]) / 100.0
            print("re-rank phases, 250 workgroups, median / p90 us after the workgroup's first stamp: " +
                  " | ".join("%s %.2f/%.2f" % (nme, np.median(t8[:, i]), np.percentile(t8[:, i], 90)) for i, nme in
                             [(1, "keys in, pass 1"), (2, "candidates chosen"), (3, "exact rows done"), (4, "pending rows staged"), (5, "pending scanned"), (6, "barrier")]))
        if rep == 5:
            rb = np.array(tb[8:16], dtype=np.float64); ft = np.array(tb[0:8], dtype=np.float64)
            print("decision loop stamps (us after its first): " + " ".join("%.2f" % ((x - rb[0]) / 100.0) for x in rb[:6]) +
                  "   [entry | neighbours + rows read | reject mask | sweeps done | prefix | word ids written]")
            print("tail stamps (us after the first): " + " ".join("%.2f" % ((x - ft[0]) / 100.0) for x in ft))
            sw = np.array(tb[16:48], dtype=np.float64).reshape(4, 8)
            print("first sweep, wave 0 (us after the decision loop's entry): " + " ".join("%.2f" % ((x - rb[0]) / 100.0) for x in sw[0]))
        if hasattr(lib, "lcd_debug_score_timing_pipe") and rep == 5:
            sb = (ctypes.c_ulonglong * (1024 * 8))()
            assert lib.lcd_debug_score_timing_pipe(sb, 1024 * 8) == 0
            s = np.frombuffer(sb, dtype=np.uint64).reshape(-1, 8).astype(np.float64)[:, :4]
            m = (s[:, 0] > s[:, 0].max() - 20000) & (s[:, 3] >= s[:, 0])
            s = (s[m] - s[m, 0].min()) / 100.0
            if hasattr(lib, "lcd_debug_b_timing"):
                # per XCD (workgroup index modulo 8) and the slowest workgroups: where the tail of launch B comes from
                bb2 = (ctypes.c_ulonglong * (2 * 4096))()
                assert lib.lcd_debug_b_timing(bb2, 2 * 4096) == 0
                b2 = np.frombuffer(bb2, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
                idx2 = np.nonzero(b2[:, 1] > b2[:, 1].max() - 20000)[0]
                t0b = b2[idx2, 0].min()
                nr2 = ((q + 1) // 2 + 7) & ~7
                sidx = idx2[idx2 >= nr2]
                dur = (b2[sidx, 1] - b2[sidx, 0]) / 100.0
                print("scoring workgroups by XCD (index mod 8): " + " ".join("%d:%.1f/%.1f" % (x, np.median(dur[sidx % 8 == x]), dur[sidx % 8 == x].max()) for x in range(8)))
                order = np.argsort(-dur)[:12]
                print("slowest scoring workgroups (launch index, us): " + " ".join("%d:%.1f" % (sidx[o], dur[o]) for o in order))
                ridx = idx2[idx2 < nr2]
                rdur = (b2[ridx, 1] - b2[ridx, 0]) / 100.0
                print("re-rank workgroups by XCD: " + " ".join("%d:%.1f/%.1f" % (x, np.median(rdur[ridx % 8 == x]), rdur[ridx % 8 == x].max()) for x in range(8)))
            print("launch B scoring phases, %d sealed-bucket workgroups (us after the first one started):" % m.sum())
            for i, nme in enumerate(["start", "lists + directory + dense rows read", "dense sums in LDS", "sparse postings done"]):
                c = s[:, i]
                print("  %-38s min %5.2f median %5.2f p90 %5.2f max %5.2f" % (nme, c.min(), np.median(c), np.percentile(c, 90), c.max()))
        eng.synchronize()
    f, tl = res[-1]
    print("%d filter waves in the last launch" % len(f))
    for i, nme in enumerate(["kernel entry", "loop entry", "loop exit", "kernel exit"]):
        c = f[:, i]
        print("  %-13s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (nme, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
    for (f, tl) in res:
        print("filter: first entry 0.00, last exit %6.2f | decision loop %6.2f -> %6.2f | registration %6.2f -> retire done %6.2f -> %6.2f" %
              (f[:, 3].max(), tl[0], tl[1], tl[4], tl[5], tl[6]))
    eng.close()


if __name__ == "__main__":
    main()
