cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_append_dev.py -x -q -m gpu > $O/append.txt 2>&1
tail -30 $O/append.txt
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_append_dev.py > $O/tests.txt 2>&1
tail -5 $O/tests.txt
