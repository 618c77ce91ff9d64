cd $GRAFT_REPO_ROOT
export PWAF_COMMIT=7348917
mkdir -p gpurun_out/final
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/final/pytest.log; cat gpurun_out/final/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -c 1500 gpurun_out/final/bench.json
PROFILE_ARGS="--config 5" bash tools/profile_round.sh r3b_c5 > gpurun_out/profile_c5b.log 2>&1
tail -30 gpurun_out/profile_c5b.log
