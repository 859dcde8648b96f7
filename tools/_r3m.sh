cd $GRAFT_REPO_ROOT
O=gpurun_out/r3x; mkdir -p $O
LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_atiming.so timeout 300 python tools/frame_a_timing.py > $O/a_timing.txt 2>&1
grep -A12 "launch A" $O/a_timing.txt; tail -3 $O/a_timing.txt
