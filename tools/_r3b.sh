cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
rocminfo | grep -E "Name:|Compute Unit|Node" | head -40 > $O/rocminfo.txt
rocm-smi > $O/smi.txt 2>&1
timeout 600 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1
tail -15 $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
