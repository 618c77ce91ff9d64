cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/c6/pytest.log
cat gpurun_out/c6/pytest.log
export PWAF_ATTR_INLINE=1
bash tools/exp_round3.sh c6 base ls2:PWAF_LIST_SHAPE=2 ls2_0:PWAF_LIST_SHAPE=2 ls1:PWAF_LIST_SHAPE=1 v1:PWAF_DEBUG_SKIP=1 v2:PWAF_DEBUG_SKIP=2 v8:PWAF_DEBUG_SKIP=8 v32:PWAF_DEBUG_SKIP=32 v128:PWAF_DEBUG_SKIP=128 v256:PWAF_DEBUG_SKIP=256 arows:PWAF_DEBUG_SKIP=0x40000 atrans:PWAF_DEBUG_SKIP=0x80000 acmp:PWAF_DEBUG_SKIP=0x100000 aall:PWAF_DEBUG_SKIP=0x1C0000 > gpurun_out/c6/exp.log 2>&1
cat gpurun_out/c6/exp.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c6adv base ls2:PWAF_LIST_SHAPE=2 ls1:PWAF_LIST_SHAPE=1 > gpurun_out/c6/exp_adv.log 2>&1
cat gpurun_out/c6/exp_adv.log
BENCH_EXTRA="--config 5" bash tools/exp_round3.sh c6c5 base ls2:PWAF_LIST_SHAPE=2 v1:PWAF_DEBUG_SKIP=1 v2:PWAF_DEBUG_SKIP=2 v8:PWAF_DEBUG_SKIP=8 v32:PWAF_DEBUG_SKIP=32 v128:PWAF_DEBUG_SKIP=128 > gpurun_out/c6/exp_c5.log 2>&1
cat gpurun_out/c6/exp_c5.log
