cd $GRAFT_REPO_ROOT
export PWAF_COMMIT=876a2b6
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/pytest10.log; cat gpurun_out/pytest10.log
bash tools/profile_round.sh r3_c3 > gpurun_out/profile_c3.log 2>&1
tail -40 gpurun_out/profile_c3.log
PROFILE_ARGS="--config 5" bash tools/profile_round.sh r3_c5 > gpurun_out/profile_c5.log 2>&1
tail -30 gpurun_out/profile_c5.log
