cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q11; mkdir -p $O
export TMPDIR=/tmp
LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_atiming.so timeout 200 python tools/frame_a_timing.py > $O/t.txt 2>&1
grep -v "^filter:" $O/t.txt | tail -30
for v in 0 5 6; do
LCD_BENCH_OPTS=strip_tiles=$v timeout 300 python bench.py --words 59000 --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/b59_$v.json 2> $O/b59_$v.err
done
for v in 0 7 8; do
LCD_BENCH_OPTS=strip_tiles=$v timeout 300 python bench.py --words 80000 --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/b80_$v.json 2> $O/b80_$v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3q11/b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["ms_per_step"]*1e3,2), round(d["roofline"]["ms"]*1e3,2), d["roofline"]["kernel"][:18], round(d["roofline_score"]["ms"]*1e3,2))
    except Exception as e: print(f, "ERR", e)
PY
