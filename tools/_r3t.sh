cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; mkdir -p $O
export TMPDIR=/tmp
LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_atiming.so timeout 300 python tools/frame_a_timing.py > $O/a_timing.txt 2>&1
grep -A3 "launch B" $O/a_timing.txt
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/bench200.json 2> $O/bench200.err
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/bench200b.json 2> $O/bench200b.err
python - <<'PY'
import json
for f in ("bench200","bench200b"):
    try:
        d=json.loads(open("gpurun_out/r3t/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["ms"], d.get("roofline_score",{}).get("ms"), d["config"]["step_ms_median"])
    except Exception as e: print(f, "ERR", e)
PY
