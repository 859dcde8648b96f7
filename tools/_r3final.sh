cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
LCD_LIB_PATH=$PWD/rtabmap_amd/liblcd_hip_atiming.so timeout 300 python tools/frame_a_timing.py > $O/r03_launch_timeline.txt 2>&1
bash tools/profile_round.sh r03 1 > /dev/null 2>&1
ls $O | head -40
