cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/c5/pytest.log
cat gpurun_out/c5/pytest.log
bash tools/exp_round3.sh c5 base inline:PWAF_ATTR_INLINE=1 early:PWAF_ATTR_EARLY=1 > gpurun_out/c5/exp.log 2>&1
cat gpurun_out/c5/exp.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c5adv base inline:PWAF_ATTR_INLINE=1 early:PWAF_ATTR_EARLY=1 > gpurun_out/c5/exp_adv.log 2>&1
cat gpurun_out/c5/exp_adv.log
BENCH_EXTRA="--config 5" bash tools/exp_round3.sh c5c5 base inline:PWAF_ATTR_INLINE=1 early:PWAF_ATTR_EARLY=1 > gpurun_out/c5/exp_c5.log 2>&1
cat gpurun_out/c5/exp_c5.log
