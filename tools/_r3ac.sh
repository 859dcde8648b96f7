cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03 1 > /dev/null 2>&1
ls gpurun_out/r03 | head -30
