cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q5; mkdir -p $O
export TMPDIR=/tmp
LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_atiming.so timeout 200 python tools/frame_a_timing.py > $O/t.txt 2>&1
grep -v "^filter:" $O/t.txt | tail -32
for rep in 1 2; do
LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip.so timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/b_$rep.json 2> $O/b_$rep.err
done
python - <<'PY'
import json,glob
r=[]
for f in sorted(glob.glob("gpurun_out/r3q5/b_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r.append((round(d["ms_per_step"]*1e3,2), round(d["roofline"]["ms"]*1e3,2), round(d["roofline_score"]["ms"]*1e3,2)))
print(r)
PY
