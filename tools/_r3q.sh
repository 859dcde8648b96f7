cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt
( time timeout 900 python bench.py --config orb_stream ) > $O/bench_orb.json 2> $O/bench_orb.err; tail -4 $O/bench_orb.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r3q/bench_orb.json").read().strip().splitlines()[-1])
    print("orb", json.dumps(d)[:2600])
except Exception as e: print("orb ERR", e)
PY
