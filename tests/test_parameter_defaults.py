"""CPU: the host mirror's classes, constructed without parameters, hold the defaults of the reference's parameter table
(corelib/include/rtabmap/core/Parameters.h; tests/golden/parameter_defaults.json is made from it by tests/golden/make_parameter_defaults.py).
No device call is made: the engine of a VWDictionaryHip is created on first use."""
import ctypes as C
import json
import os

import numpy as np


def test_mirror_defaults_are_the_reference_defaults():
    from rtabmap_amd import vwdictionary as V
    L = V.lib()
    L.hparams_defaults.argtypes = [C.POINTER(C.c_double), C.c_int]
    out = (C.c_double * 64)()
    n_lc = L.hparams_defaults(out, 64)
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "parameter_defaults.json")))
    f32 = lambda s: float(np.float32(float(s)))
    truth = lambda s: 1.0 if s == "true" else 0.0
    assert out[0] == f32(ref["Kp/NndrRatio"]["default"])                       # the mirror keeps floats, as the reference's members are
    assert out[1] == truth(ref["Kp/IncrementalDictionary"]["default"])
    assert out[2] == truth(ref["Kp/NewWordsComparedTogether"]["default"])
    assert out[3] == float(int(ref["Mem/STMSize"]["default"]))
    assert out[4] == f32(ref["Rtabmap/LoopThr"]["default"])
    assert out[5] == f32(ref["Rtabmap/LoopRatio"]["default"])
    assert out[6] == f32(ref["Bayes/VirtualPlacePriorThr"]["default"])
    assert out[7] == truth(ref["Bayes/FullPredictionUpdate"]["default"])
    lc = [f32(x) for x in ref["Bayes/PredictionLC"]["default"].split()]       # each value goes through a float (BayesFilter.cpp:100-105)
    assert n_lc == len(lc) == 18 and [out[8 + i] for i in range(n_lc)] == lc
    assert ref["Kp/TfIdfLikelihoodUsed"]["default"] == "true"                  # the branch of computeLikelihood this engine implements
    assert ref["Kp/DictionaryPath"]["default"] == ""                           # incremental dictionary unless a path is given
    assert ref["Kp/NNStrategy"]["default"] == "1"                              # the reference's default is its kd-tree: strategy 5 is opt-in
