cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c21adv
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c21adv rowsadv:PWAF_BENCH_RETUNE_ROWS_ADV=1 rowsadv_never:PWAF_BENCH_RETUNE_ROWS_ADV=1,PWAF_LSCAN_ASYNC=1 > gpurun_out/c21adv/exp.log 2>&1
cat gpurun_out/c21adv/exp.log
