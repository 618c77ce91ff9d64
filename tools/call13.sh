cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c13 gpurun_out/c13adv gpurun_out/c13c5
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/pytest13.log; cat gpurun_out/pytest13.log
bash tools/exp_round3.sh c13 mix s1only:PWAF_STRIDE2_FIELDS=0 mixnoheads:PWAF_FILTER_DEBUG_SKIP=4 mixnolook:PWAF_FILTER_DEBUG_SKIP=1 > gpurun_out/c13/exp.log 2>&1
cat gpurun_out/c13/exp.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c13adv base > gpurun_out/c13adv/exp.log 2>&1
cat gpurun_out/c13adv/exp.log
BENCH_EXTRA="--config 5" bash tools/exp_round3.sh c13c5 base > gpurun_out/c13c5/exp.log 2>&1
cat gpurun_out/c13c5/exp.log
