cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c9
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/c9/pytest.log
cat gpurun_out/c9/pytest.log
bash tools/exp_round3.sh c9 base inline:PWAF_PLACEMENT=1 ls0:PWAF_LIST_SHAPE=0 ls22:PWAF_LIST_SHAPE=34 > gpurun_out/c9/exp.log 2>&1
cat gpurun_out/c9/exp.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c9adv base inline:PWAF_PLACEMENT=1 ls0:PWAF_LIST_SHAPE=0 ls22:PWAF_LIST_SHAPE=34 > gpurun_out/c9/exp_adv.log 2>&1
cat gpurun_out/c9/exp_adv.log
BENCH_EXTRA="--config 5" bash tools/exp_round3.sh c9c5 base inline:PWAF_PLACEMENT=1 > gpurun_out/c9/exp_c5.log 2>&1
cat gpurun_out/c9/exp_c5.log
BENCH_EXTRA="--config 5 --adversarial" bash tools/exp_round3.sh c9c5adv base inline:PWAF_PLACEMENT=1 ls0:PWAF_LIST_SHAPE=0 ls22:PWAF_LIST_SHAPE=34 > gpurun_out/c9/exp_c5adv.log 2>&1
cat gpurun_out/c9/exp_c5adv.log
