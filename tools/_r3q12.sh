cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q12; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for w in 49000 59000 80000 112000; do
timeout 300 python bench.py --words $w --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/b_$w.json 2> $O/b_$w.err
done
timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/b_extras.json 2> $O/b_extras.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3q12/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["config"]
        print(f.split("/")[-1], round(d["ms_per_step"]*1e3,2), round(d["roofline"]["ms"]*1e3,2), d["roofline"]["kernel"][:18], round(d["roofline_score"]["ms"]*1e3,2), {k:round(c[k]*1e3,1) for k in c if k.startswith("with_") and k.endswith("per_step")})
    except Exception as e: print(f, "ERR", e)
PY
