cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1
tail -5 $O/tests.txt
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench200.json 2> $O/bench200.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench20.json 2> $O/bench20.err
python - <<'PY'
import json
for f in ("bench200","bench20"):
    try:
        d=json.loads(open("gpurun_out/r3j/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["ms"], d.get("roofline_score",{}).get("ms"), {k:v for k,v in d["config"].items() if "ms" in k})
    except Exception as e: print(f, "ERR", e)
PY
