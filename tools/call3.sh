cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/c3/pytest.log
cat gpurun_out/c3/pytest.log
bash tools/exp_round3.sh c3 base inline:PWAF_ATTR_INLINE=1 inline_ls2:PWAF_ATTR_INLINE=1,PWAF_LIST_SHAPE=2 ls2:PWAF_LIST_SHAPE=2 early:PWAF_ATTR_EARLY=1 > gpurun_out/c3/exp.log 2>&1
cat gpurun_out/c3/exp.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c3adv base inline:PWAF_ATTR_INLINE=1 > gpurun_out/c3/exp_adv.log 2>&1
cat gpurun_out/c3/exp_adv.log
BENCH_EXTRA="--config 5" bash tools/exp_round3.sh c3c5 base inline:PWAF_ATTR_INLINE=1 > gpurun_out/c3/exp_c5.log 2>&1
cat gpurun_out/c3/exp_c5.log
