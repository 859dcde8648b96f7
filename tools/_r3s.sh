cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s; mkdir -p $O
timeout 600 python -m pytest tests/test_sharded.py tests/test_abi.py -x -q > $O/tests.txt 2>&1; tail -8 $O/tests.txt
