cd $GRAFT_REPO_ROOT
export PWAF_COMMIT=de67b5d
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/pytest19.log; cat gpurun_out/pytest19.log
bash tools/profile_round.sh r3b_c3 > gpurun_out/profile_c3b.log 2>&1
tail -45 gpurun_out/profile_c3b.log
