cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q13; mkdir -p $O
export TMPDIR=/tmp
APPEND=1 LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_atiming.so timeout 200 python tools/frame_a_timing.py > $O/t_app.txt 2>&1
grep -v "^filter:" $O/t_app.txt | grep -A8 "launch A:"; grep "decision loop stamps\|tail stamps" $O/t_app.txt
grep "^filter:" $O/t_app.txt | tail -3
