cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/c2/pytest.log
cat gpurun_out/c2/pytest.log
bash tools/exp_round3.sh c2 base ls1:PWAF_LIST_SHAPE=1 ls2:PWAF_LIST_SHAPE=2 ls3:PWAF_LIST_SHAPE=3 noattr:PWAF_SKIP_ATTR=1 noipres:PWAF_SKIP_IPRES=1 norows:PWAF_DEBUG_SKIP=0x40000 notrans:PWAF_DEBUG_SKIP=0x80000 nocmp:PWAF_DEBUG_SKIP=0x100000 > gpurun_out/c2/exp.log 2>&1
cat gpurun_out/c2/exp.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c2adv base ls1:PWAF_LIST_SHAPE=1 ls2:PWAF_LIST_SHAPE=2 > gpurun_out/c2/exp_adv.log 2>&1
cat gpurun_out/c2/exp_adv.log
BENCH_EXTRA="--config 5" bash tools/exp_round3.sh c2c5 base ls1:PWAF_LIST_SHAPE=1 ls3:PWAF_LIST_SHAPE=3 > gpurun_out/c2/exp_c5.log 2>&1
cat gpurun_out/c2/exp_c5.log
