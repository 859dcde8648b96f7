"""Regenerates tests/golden/parameter_defaults.json from the reference's parameter table.

Run in the authoring container (needs /root/reference):  python tests/golden/make_parameter_defaults.py

Source: corelib/include/rtabmap/core/Parameters.h -- the RTABMAP_PARAM / RTABMAP_PARAM_STR lines of the keys the hot path reads
(Kp/* :243-266, Mem/STMSize :214, Rtabmap/LoopThr, LoopRatio :197-198, Bayes/* :362-364).  Only the default values are stored."""
import json
import os
import re

REF = os.environ.get("LCD_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ["Kp/NndrRatio", "Kp/IncrementalDictionary", "Kp/NewWordsComparedTogether", "Kp/NNStrategy", "Kp/TfIdfLikelihoodUsed", "Kp/DictionaryPath",
        "Mem/STMSize", "Rtabmap/LoopThr", "Rtabmap/LoopRatio", "Bayes/VirtualPlacePriorThr", "Bayes/PredictionLC", "Bayes/FullPredictionUpdate"]


def main():
    txt = open(os.path.join(REF, "corelib/include/rtabmap/core/Parameters.h")).read()
    out = {}
    for key in KEYS:
        group, name = key.split("/")
        m = re.search(r"RTABMAP_PARAM\(\s*%s\s*,\s*%s\s*,\s*([^,]+?)\s*,\s*([^,]+?)\s*," % (group, name), txt)
        if m:
            typ, val = m.group(1).strip(), m.group(2).strip()
            out[key] = {"type": typ, "default": val}
            continue
        m = re.search(r'RTABMAP_PARAM_STR\(\s*%s\s*,\s*%s\s*,\s*"([^"]*)"' % (group, name), txt)
        if not m:
            raise SystemExit("no default found for " + key)
        out[key] = {"type": "string", "default": m.group(1)}
    with open(os.path.join(HERE, "parameter_defaults.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
