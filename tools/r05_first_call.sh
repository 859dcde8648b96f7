#!/bin/bash
# The first GPU call of the next round: what round 4 prepared without a GPU, verified and timed in one pass (~10-12 min of box time).
#   (here)   python tools/build_variant.py apprr -DLCD_APPEND_FROM_RERANK
#   (here)   python tools/build_variant.py hchain -DLCD_HAMMING_CHAIN
#   (here)   gpurun --timeout 1200 -- 'bash tools/r05_first_call.sh'
# Writes gpurun_out/r05a/: the GPU suite on the product library, the append / frame-stream suites on the variant library, a same-box A/B
# of the two (200 steps, three alternating runs each, headline and 10^6 signatures); the Hamming-scan variant (bit counts accumulated in the
# instruction: 94 -> 82 VALU per four rows) on the exact-scan suites and against the product scan at 200 000 ORB words.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/r05a
mkdir -p $O
VAR=$ROOT/rtabmap_amd/liblcd_hip_apprr.so

# 1. the product library: the whole GPU suite (the round-end check, early)
timeout 400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_product.log 2>&1
tail -2 $O/pytest_product.log

# 1b. tests written in round 4 without a GPU (skipped by `-m gpu` until they have passed once: then remove their skipif)
LCD_RUN_UNVERIFIED=1 timeout 200 python -m pytest tests/test_gpu_db_load.py -x -q -p no:cacheprovider > $O/pytest_unverified.log 2>&1
tail -2 $O/pytest_unverified.log

# 2. the variant (re-rank workgroups write the rows of the deferred append): every suite that appends on the device
if [ -f $VAR ]; then
    LCD_LIB_PATH=$VAR timeout 300 python -m pytest tests/test_gpu_append_dev.py tests/test_gpu_frame_stream.py tests/test_gpu_fuzz.py tests/test_gpu_quantize.py \
        -x -q -p no:cacheprovider > $O/pytest_apprr.log 2>&1
    tail -2 $O/pytest_apprr.log
    # 3. same-box A/B, alternating: the headline step and the 10^6-signature memory (where the product library pays a third launch)
    B="python bench.py --warmup 20 --steps 200 --no-cpu-baseline --no-extras --no-pmc"
    for i in 1 2 3; do
        timeout 120 $B > $O/h_prod_$i.json 2>/dev/null
        LCD_LIB_PATH=$VAR timeout 120 $B > $O/h_apprr_$i.json 2>/dev/null
    done
    for i in 1 2; do
        timeout 200 $B --signatures 1000000 > $O/m_prod_$i.json 2>/dev/null
        LCD_LIB_PATH=$VAR timeout 200 $B --signatures 1000000 > $O/m_apprr_$i.json 2>/dev/null
    done
    python - $O <<'P'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/[hm]_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(f.split("/")[-1], "ms/step %.4f" % d["ms_per_step"], "median %.4f" % c.get("step_ms_median", 0.0),
              "A %.4f" % d.get("roofline", {}).get("ms", 0.0), "B %.4f" % (d.get("roofline_score") or {}).get("ms", 0.0),
              "parity", (d.get("parity") or {}).get("ok"))
    except Exception as e:
        print(f, "ERR", e)
P
fi

# 4. the Hamming scan with the bit counts accumulated in the instruction (knn2_kernels.hip, LCD_HAMMING_CHAIN): bit-exact suites, then the scan's
#    own time at 200 000 words x 500 descriptors (tools/bench_orb.py prints the kernel's HIP-event time), alternating
HV=$ROOT/rtabmap_amd/liblcd_hip_hchain.so
if [ -f $HV ]; then
    LCD_LIB_PATH=$HV timeout 300 python -m pytest tests/test_gpu_knn.py tests/test_gpu_quantize.py tests/test_gpu_fuzz.py -x -q -p no:cacheprovider > $O/pytest_hchain.log 2>&1
    tail -2 $O/pytest_hchain.log
    for i in 1 2; do
        timeout 120 python tools/bench_orb.py > $O/orb_prod_$i.json 2>/dev/null
        LCD_LIB_PATH=$HV timeout 120 python tools/bench_orb.py > $O/orb_hchain_$i.json 2>/dev/null
    done
    tail -n 1 $O/orb_prod_*.json $O/orb_hchain_*.json
fi
