cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c15 gpurun_out/c15adv gpurun_out/c15c5
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest15.log; cat gpurun_out/pytest15.log
bash tools/exp_round3.sh c15 local whole:PWAF_WHOLE_WALKS=1 lockstep:PWAF_LOCKSTEP_WALKS=1 > gpurun_out/c15/exp.log 2>&1
cat gpurun_out/c15/exp.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c15adv local whole:PWAF_WHOLE_WALKS=1 lockstep:PWAF_LOCKSTEP_WALKS=1 pathS1:PWAF_STRIDE2_FIELDS=0x11 > gpurun_out/c15adv/exp.log 2>&1
cat gpurun_out/c15adv/exp.log
BENCH_EXTRA="--config 5" bash tools/exp_round3.sh c15c5 local > gpurun_out/c15c5/exp.log 2>&1
cat gpurun_out/c15c5/exp.log
BENCH_EXTRA="--config 5 --adversarial" bash tools/exp_round3.sh c15c5 localadv lockstepadv:PWAF_LOCKSTEP_WALKS=1 > gpurun_out/c15c5/expadv.log 2>&1
cat gpurun_out/c15c5/expadv.log
