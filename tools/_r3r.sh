cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python bench.py --config replay --signatures 30000 ) > $O/replay_30k.json 2> $O/replay_30k.err; tail -5 $O/replay_30k.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r3r/replay_30k.json").read().strip().splitlines()[-1])
    print(json.dumps(d)[:2400])
except Exception as e: print("ERR", e)
PY
