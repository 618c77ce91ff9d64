cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c14 gpurun_out/c14adv gpurun_out/c14c5
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/pytest14.log; cat gpurun_out/pytest14.log
bash tools/exp_round3.sh c14 local whole:PWAF_WHOLE_WALKS=1 > gpurun_out/c14/exp.log 2>&1
cat gpurun_out/c14/exp.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c14adv local whole:PWAF_WHOLE_WALKS=1 pathS1:PWAF_STRIDE2_FIELDS=0x11 > gpurun_out/c14adv/exp.log 2>&1
cat gpurun_out/c14adv/exp.log
BENCH_EXTRA="--config 5" bash tools/exp_round3.sh c14c5 local > gpurun_out/c14c5/exp.log 2>&1
cat gpurun_out/c14c5/exp.log
BENCH_EXTRA="--config 5 --adversarial" bash tools/exp_round3.sh c14c5 localadv wholeadv:PWAF_WHOLE_WALKS=1 > gpurun_out/c14c5/expadv.log 2>&1
cat gpurun_out/c14c5/expadv.log
