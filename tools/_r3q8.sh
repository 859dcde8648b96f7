cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q8; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for v in prod nowrite; do
L=$PWD/rtabmap_amd/liblcd_hip.so
if [ $v = nowrite ]; then L=$PWD/rtabmap_amd/liblcd_hip_nowrite.so; fi
LCD_LIB_PATH=$L timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err
done; done
python - <<'PY'
import json,glob
for v in ("prod","nowrite"):
    r=[]
    for f in sorted(glob.glob("gpurun_out/r3q8/b_%s_*.json"%v)):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r.append((round(d["ms_per_step"]*1e3,2), round(d["roofline"]["ms"]*1e3,2), d["roofline"]["samples"], round(d["roofline_score"]["ms"]*1e3,2)))
        except Exception as e: r.append(("ERR",str(e)))
    print(v, r)
PY
tail -2 $O/b_nowrite_1.err
