cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q3; mkdir -p $O
export TMPDIR=/tmp
for v in 0 8 3; do
LCD_BENCH_OPTS=strip_tiles=$v LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_atiming.so timeout 200 python tools/frame_a_timing.py > $O/t_$v.txt 2>&1
echo "== strip_tiles=$v"; grep -v "^filter:" $O/t_$v.txt | tail -16; grep "^filter:" $O/t_$v.txt | tail -2
done
