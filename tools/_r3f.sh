cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "frame or stream or quantize or knn" > $O/tests.txt 2>&1
tail -5 $O/tests.txt
LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_atiming.so timeout 300 python tools/frame_a_timing.py > $O/a_timing.txt 2>&1
grep -A12 "launch A" $O/a_timing.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench200.json 2> $O/bench200.err
python - <<'PY'
import json
for f in ("bench20","bench200"):
    try:
        d=json.loads(open("gpurun_out/r3g/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["ms"], d.get("roofline_score",{}).get("ms"), {k:v for k,v in d["config"].items() if "ms" in k})
    except Exception as e: print(f, "ERR", e)
PY
