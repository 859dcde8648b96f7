cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -s > $O/tests.txt 2>&1
tail -12 $O/tests.txt; grep "temporary dictionary" $O/tests.txt
