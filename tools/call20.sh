cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c20 gpurun_out/c20adv gpurun_out/c20c5
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/pytest20.log; cat gpurun_out/pytest20.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c20adv auto never:PWAF_LSCAN_ASYNC=1 always:PWAF_LSCAN_ASYNC=2 > gpurun_out/c20adv/exp.log 2>&1
cat gpurun_out/c20adv/exp.log
bash tools/exp_round3.sh c20 auto always:PWAF_LSCAN_ASYNC=2 > gpurun_out/c20/exp.log 2>&1
cat gpurun_out/c20/exp.log
BENCH_EXTRA="--config 5 --adversarial" bash tools/exp_round3.sh c20c5 autoadv neveradv:PWAF_LSCAN_ASYNC=1 > gpurun_out/c20c5/expadv.log 2>&1
cat gpurun_out/c20c5/expadv.log
