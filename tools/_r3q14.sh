cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q14; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
