cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q4; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "frame or append or stream" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for v in 0 6; do
LCD_BENCH_OPTS=strip_tiles=$v LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_atiming.so timeout 200 python tools/frame_a_timing.py > $O/t_$v.txt 2>&1
echo "== strip_tiles=$v"; grep -A6 "launch A" $O/t_$v.txt; grep -A5 "filter waves" $O/t_$v.txt
done
for rep in 1 2; do
for v in prev 0 6 7; do
if [ $v = prev ]; then L=$PWD/rtabmap_amd/liblcd_hip_prev.so; OPT=""; else L=$PWD/rtabmap_amd/liblcd_hip.so; OPT="strip_tiles=$v"; fi
LCD_BENCH_OPTS=$OPT LCD_LIB_PATH=$L timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err
done; done
python - <<'PY'
import json,glob
for v in ("prev","0","6","7"):
    r=[]
    for f in sorted(glob.glob("gpurun_out/r3q4/b_%s_*.json"%v)):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r.append((round(d["ms_per_step"]*1e3,2), round(d["roofline"]["ms"]*1e3,2), round(d["roofline_score"]["ms"]*1e3,2)))
        except Exception as e: r.append(("ERR",str(e)))
    print(v, r)
PY
