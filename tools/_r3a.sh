cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench200.json 2> $O/bench200.err
PIPE=1 LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_tailtiming.so timeout 300 python tools/tail_timing.py > $O/tail_pipe.txt 2>&1
tail -3 $O/tail_pipe.txt
python - <<'PY'
import json
for f in ("bench20","bench200"):
    try:
        d=json.loads(open("gpurun_out/r3a/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["ms"], d.get("roofline_score",{}).get("ms"), {k:v for k,v in d["config"].items() if "ms" in k})
    except Exception as e: print(f, "ERR", e)
PY
