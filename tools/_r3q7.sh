cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q7; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for v in prev new new20; do
L=$PWD/rtabmap_amd/liblcd_hip.so; N=""
if [ $v = prev ]; then L=$PWD/rtabmap_amd/liblcd_hip_prev.so; fi
if [ $v = new20 ]; then export LCD_BENCH_PROF_N=20; else unset LCD_BENCH_PROF_N; fi
LCD_LIB_PATH=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err
done; done
unset LCD_BENCH_PROF_N
LCD_BENCH_PROF_N=300 timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/b_all_1.json 2> $O/b_all_1.err
timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/b_300_1.json 2> $O/b_300_1.err
python - <<'PY'
import json,glob
for v in ("prev","new","new20","all","300"):
    r=[]
    for f in sorted(glob.glob("gpurun_out/r3q7/b_%s_*.json"%v)):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r.append((round(d["ms_per_step"]*1e3,2), round(d["roofline"]["ms"]*1e3,2), d["roofline"]["samples"], round(d["roofline_score"]["ms"]*1e3,2)))
        except Exception as e: r.append(("ERR",str(e)))
    print(v, r)
PY
