cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q1; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "frame or append or stream or bayes" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for rep in 1 2; do
for v in prev prod; do
if [ $v = prod ]; then L=$PWD/rtabmap_amd/liblcd_hip.so; else L=$PWD/rtabmap_amd/liblcd_hip_$v.so; fi
LCD_LIB_PATH=$L timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err
done; done
python - <<'PY'
import json,glob
for v in ("prev","prod"):
    r=[]
    for f in sorted(glob.glob("gpurun_out/r3q1/b_%s_*.json"%v)):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r.append((round(d["ms_per_step"]*1e3,2), round(d["roofline"]["ms"]*1e3,2), round(d["roofline_score"]["ms"]*1e3,2)))
        except Exception as e: r.append(("ERR",str(e)))
    print(v, r)
PY
tail -3 $O/b_prod_1.err
