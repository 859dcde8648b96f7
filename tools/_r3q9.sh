cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q9; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_atiming.so timeout 200 python tools/frame_a_timing.py > $O/t.txt 2>&1
grep -A9 "launch A:" $O/t.txt | grep -v "^filter:"
for rep in 1 2; do
timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/b_300_$rep.json 2> $O/b_300_$rep.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b_20_$rep.json 2> $O/b_20_$rep.err
done
python - <<'PY'
import json,glob
for v in ("300","20"):
    r=[]
    for f in sorted(glob.glob("gpurun_out/r3q9/b_%s_*.json"%v)):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r.append((round(d["ms_per_step"]*1e3,2), round(d["roofline"]["ms"]*1e3,2), d["roofline"]["samples"], round(d["roofline_score"]["ms"]*1e3,2)))
        except Exception as e: r.append(("ERR",str(e)))
    print(v, r)
PY
