"""Pins the statistics of the restated Rtabmap::adjustLikelihood (oracle/lcd_oracle.cpp; Rtabmap.cpp:5691-5760) against the reference's
OWN uMean / uVariance templates (utilite UMath.h:419-432, 512-526) compiled in place (oracle/_ref/librtflann_ref.so): the float
accumulation order of those two helpers decides the bits of every adjusted likelihood.  The statements around them are replayed here in
numpy float32, one rounding per C operation.  Also: the host mirror's uStr2Float (VWDictionaryHip.h) against the reference's
(UConversion.cpp), which reads Kp/NndrRatio, Rtabmap/LoopThr and the text dictionary.  CPU only."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.have_ref():
        try:
            oracle.build(ref=True)
        except Exception:
            pass
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/librtflann_ref.so not built and /root/reference absent")
    r = oracle.ref()
    if not hasattr(r, "ref_umean_list"):
        try:
            oracle.build(ref=True)
        except Exception:
            pass
        pytest.skip("oracle/_ref/librtflann_ref.so predates ref_umean_list: rebuilt for the next run (make -C oracle ref)")
    return r


def _adjust_with_reference_statistics(ref, L, ratio):
    """Rtabmap::adjustLikelihood statement by statement, mean and variance from the reference's templates."""
    f = np.float32
    L = L.astype(np.float32).copy()
    values = np.ascontiguousarray(L[1:][L[1:] > 0])
    p = values.ctypes.data_as(C.c_void_p)
    mean = f(ref.ref_umean_list(p, values.size))
    var = f(ref.ref_uvariance_list(p, values.size, mean))
    std = f(np.sqrt(var))
    eps, mx = f(0.0001), f(0.0)
    out = L.copy()
    for i in range(1, L.size):
        v = L[i]
        out[i] = f(1.0)
        if v > f(mean + std):
            if ratio == 0 and mean:
                out[i] = f(f(v - f(std - eps)) / mean)
            elif ratio != 0 and std:
                out[i] = f(f(v - mean) / std)
        if v > mx:
            mx = v
    if ratio == 0 and std > eps and mx:
        out[0] = f(f(mean / std) + f(1.0))
    elif ratio != 0 and mx > mean:
        out[0] = f(f(std / f(mx - mean)) + f(1.0))
    else:
        out[0] = f(2.0)
    return out


@pytest.mark.parametrize("ratio", [0.0, 0.5])
def test_adjust_likelihood_statistics_are_the_reference_templates(oracle, ref, ratio):
    rng = np.random.default_rng(17)
    cases = []
    for n in (2, 3, 5, 17, 84, 500, 4001):
        for kind in range(4):
            L = rng.random(n).astype(np.float32) * np.float32(0.02)
            if kind == 1:
                L[rng.random(n) < 0.7] = 0.0                     # most places share no word with the frame
            if kind == 2:
                L[1:] = np.float32(0.0125)                       # all equal: variance 0
            if kind == 3:
                L[rng.random(n) < 0.3] *= np.float32(-1.0)       # negative idf terms (N < nw)
                L[rng.integers(1, n)] = np.float32(0.9)          # one outstanding place
            cases.append(L)
    cases.append(np.zeros(40, np.float32))                      # nothing positive: mean 0, the virtual place gets 2
    cases.append(np.array([0.0, 0.3], np.float32))              # one value: uVariance of a single element is 0
    for L in cases:
        got = oracle.adjust_likelihood(L, ratio)
        exp = _adjust_with_reference_statistics(ref, L, ratio)
        np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32), err_msg="n=%d" % L.size)


def test_mirror_number_parser_is_the_reference_one(ref):
    from rtabmap_amd import vwdictionary as V
    L = V.lib()
    L.hutil_str2float.restype = C.c_float
    L.hutil_str2float.argtypes = [C.c_char_p]
    for s in ["0.8", "0,8", "1e-3", "1,5e2", "7", "-2,25", "  3.5", "3.5  ", "abc", "", "1.5f", "1.2.3", "0.11", ".5", "5.", "+4", "1e", "0x10",
              "0.10000000149", "123456789.125", "1e39", "-1e-46"]:
        a, b = L.hutil_str2float(s.encode()), ref.ref_ustr2float(s.encode())
        assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32) or (np.isnan(a) and np.isnan(b)), (s, a, b)


def test_database_version_order_is_the_reference_one(ref):
    """DbLoaderHip::versionCmp (the schema switches of the database reader) against the reference's uStrNumCmp (UStl.h:717-790) on
    every pair of versions the driver compares or a database can carry"""
    if not hasattr(ref, "ref_ustrnumcmp"):
        pytest.skip("oracle/_ref/librtflann_ref.so predates ref_ustrnumcmp (make -C oracle ref)")
    from rtabmap_amd import vwdictionary as V
    L = V.lib()
    L.hdb_version_cmp.argtypes = [C.c_char_p, C.c_char_p]
    versions = ["0.0.0", "0.9.0", "0.10.0", "0.11.2", "0.11.10", "0.11.11", "0.11.12", "0.12.0", "0.12.9", "0.13.0", "0.14.0", "0.17.3", "0.18.0",
                "0.20.23", "0.21.4", "0.23.0", "1.0.0", "0.21", "0.9"]
    sign = lambda x: (x > 0) - (x < 0)
    for a in versions:
        for b in versions:
            assert sign(L.hdb_version_cmp(a.encode(), b.encode())) == sign(ref.ref_ustrnumcmp(a.encode(), b.encode())), (a, b)


AWKWARD = """WordID Descriptors...4
7 0.5 1,25 -3e-2 4
8  1  2 3   4
9 1 2 3
7 9 9 9 9
12 0.1234564 0.1234565 1e-7 123456.789 
abc 1 2 3 4
10\t1 2 3 4
5 1 2 3 4 5
3 .5 5. +4 -0
11 1 2 3 4\r
"""


def test_text_dictionary_tokenisation_is_the_reference_one(oracle, ref, tmp_path, capfd):
    """The restated loaders of the fixed text dictionary (oracle: lcd_oracle.cpp, mirror: VWDictionaryHip::setFixedDictionary) against
    the reference's reader statements run with its OWN uSplitNumChar / uSplit / uStr2Float (oracle/rtflann_ref.cpp), on a file with
    repeated spaces, decimal commas, lines of the wrong length, a repeated id, a non-numeric id, a tab and a carriage return: the same
    words in, the same text out (exportDictionary's %f).  The mirror indexes nothing here (no device): its host maps are what is exported."""
    if not hasattr(ref, "ref_dictionary_text_roundtrip"):
        pytest.skip("oracle/_ref/librtflann_ref.so predates ref_dictionary_text_roundtrip (make -C oracle ref)")
    from rtabmap_amd.vwdictionary import VWDictionaryHip
    src = tmp_path / "awkward.txt"
    src.write_text(AWKWARD)
    want = tmp_path / "reference.txt"
    n = ref.ref_dictionary_text_roundtrip(str(src).encode(), str(want).encode())
    assert n == 6                                                           # ids 0 ("abc"), 3, 7, 8, 11, 12
    text = want.read_text()
    assert text.splitlines()[0] == "WordID Descriptors...4" and [l.split()[0] for l in text.splitlines()[1:]] == ["0", "3", "7", "8", "11", "12"]
    assert text.splitlines()[3] == "7 0.500000 1.250000 -0.030000 4.000000 "  # the first line of id 7 stays; the decimal comma is read
    o = oracle.OracleVWDictionary(strategy=oracle.kNNBruteForce, incremental=False)
    assert o.load_fixed_text(str(src)) == n + 1 and o.visual_words == n          # (the oracle's return value counts the accepted LINES: id 7 twice)
    o.export_text(None, str(tmp_path / "oracle.txt"))
    assert (tmp_path / "oracle.txt").read_text() == text
    h = VWDictionaryHip(incremental=False, dictionary_path=str(src))         # (logs that no device engine can be created: expected here)
    assert h.visual_words == n
    h.export_text(None, str(tmp_path / "mirror.txt"))
    assert (tmp_path / "mirror.txt").read_text() == text
    h.close()
    capfd.readouterr()
