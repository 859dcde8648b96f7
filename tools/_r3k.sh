cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_append_dev.py tests/test_gpu_frame_stream.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench200.json 2> $O/bench200.err
python - <<'PY'
import json
for f in ("bench200",):
    try:
        d=json.loads(open("gpurun_out/r3n/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["ms"], d.get("roofline_score",{}).get("ms"), {k:v for k,v in d["config"].items() if "ms" in k})
    except Exception as e: print(f, "ERR", e)
PY
