"""Regenerates tests/golden/loopclosure2010.npz from the reference's MATLAB known-answer tests.

Run in the authoring container (needs /root/reference):  python tests/golden/make_golden.py

Sources (reference archive/2010-LoopClosure/Tests/, "the same tests than the ones used in the c++ library corelib"):
  090306-3_db-Signatures.txt   83 signatures incl. the virtual place -1   (format of Memory::dumpSignatures)
  090306-3_db-Dictionary.txt   4554 words -> referencing signature ids    (format of VWDictionary::exportDictionary)
  TestComputeLikelihood.m:16   expected words of the query signature (id 82)
  TestComputeLikelihood.m:27   expected floor(likelihood*1000), 83 entries
  TestUpdateCommonSignature.m:24  expected sorted common words of the virtual place
Only data is stored (integer matrices + the golden vectors); no reference source is copied.
"""
import os
import re
import numpy as np

REF = os.environ.get("LCD_REFERENCE", "/root/reference")
T = os.path.join(REF, "archive/2010-LoopClosure/Tests")
HERE = os.path.dirname(os.path.abspath(__file__))


def dlmread(path):
    rows = []
    with open(path) as f:
        next(f)                       # header line (dlmread(..., ' ', 1, 0))
        for line in f:
            tok = line.split()
            if tok:
                rows.append([int(t) for t in tok])
    w = max(len(r) for r in rows)
    m = np.zeros((len(rows), w), np.int32)
    for i, r in enumerate(rows):
        m[i, :len(r)] = r
    return m


def vectors_in(path):
    """All bracketed integer row vectors '[a,b,c;]' of an .m file, in order of appearance."""
    txt = open(path).read()
    return [np.array([int(x) for x in m.group(1).split(",")], np.int32)
            for m in re.finditer(r"\[([-0-9,]+);\]", txt)]


def main():
    sig = dlmread(os.path.join(T, "090306-3_db-Signatures.txt"))
    dic = dlmread(os.path.join(T, "090306-3_db-Dictionary.txt"))
    v = vectors_in(os.path.join(T, "TestComputeLikelihood.m"))
    assert len(v) == 2, len(v)
    query_row, golden_likelihood = v
    cw = vectors_in(os.path.join(T, "TestUpdateCommonSignature.m"))
    assert len(cw) == 1
    out = os.path.join(HERE, "loopclosure2010.npz")
    np.savez_compressed(out, signatures=sig, dictionary=dic, query_row=query_row,
                        golden_likelihood_floor1000=golden_likelihood, golden_common_words=cw[0])
    print("wrote", out, sig.shape, dic.shape, query_row.shape, golden_likelihood.shape, cw[0].shape)


if __name__ == "__main__":
    main()
