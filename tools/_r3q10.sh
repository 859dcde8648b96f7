cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q10; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "frame or append or stream or bayes" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for rep in 1 2 3; do
for v in prev prod; do
L=$PWD/rtabmap_amd/liblcd_hip.so
if [ $v = prev ]; then L=$PWD/rtabmap_amd/liblcd_hip_prev.so; fi
LCD_LIB_PATH=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b_${v}20_$rep.json 2> $O/b_${v}20_$rep.err
done; done
for rep in 1 2; do
timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/b_300_$rep.json 2> $O/b_300_$rep.err
done
python - <<'PY'
import json,glob
for v in ("prev20","prod20","300"):
    r=[]
    for f in sorted(glob.glob("gpurun_out/r3q10/b_%s_*.json"%v)):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["config"]; r.append((round(d["ms_per_step"]*1e3,2), round(d["roofline"]["ms"]*1e3,2), round(c["host_enqueue_ms_per_step"]*1e3,1), round(c["host_ms_inside_lcd_frame_dev"]*1e3,1)))
        except Exception as e: r.append(("ERR",str(e)))
    print(v, r)
PY
