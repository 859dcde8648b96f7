cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ab; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
for rep in 1 2 3; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b20_$rep.json 2> $O/b20_$rep.err
done
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/b200.json 2> $O/b200.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3ab/b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"]*1e3,2), d["value"], d["roofline"]["ms"], d["roofline"]["samples"], d["roofline_score"]["ms"], d["config"]["step_ms_median"], d["config"].get("with_update_ms_per_step"), d["config"].get("with_append_ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
