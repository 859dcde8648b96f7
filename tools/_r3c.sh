cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "frame or stream or quantize or knn" > $O/tests.txt 2>&1
tail -15 $O/tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench200.json 2> $O/bench200.err
python - <<'PY'
import json
for f in ("bench20","bench200"):
    try:
        d=json.loads(open("gpurun_out/r3e/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["ms"], d.get("roofline_score",{}).get("ms"), {k:v for k,v in d["config"].items() if "ms" in k}, d.get("parity"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $O/bench20.err
