cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c1/pytest.log
timeout 900 python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err
tail -c 600 gpurun_out/c1/bench.err
bash tools/exp_round3.sh c1 base fnolook:PWAF_FILTER_DEBUG_SKIP=1 fnoload:PWAF_FILTER_DEBUG_SKIP=2 fnoheads:PWAF_FILTER_DEBUG_SKIP=4 fnone:PWAF_FILTER_DEBUG_SKIP=7 noattr:PWAF_SKIP_ATTR=1 > gpurun_out/c1/exp.log 2>&1
cat gpurun_out/c1/pytest.log; cat gpurun_out/c1/exp.log
