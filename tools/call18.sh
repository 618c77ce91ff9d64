cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c18
bash tools/exp_round3.sh c18 base place0:PWAF_PLACEMENT=0 place3:PWAF_PLACEMENT=3 place1:PWAF_PLACEMENT=1 > gpurun_out/c18/exp.log 2>&1
cat gpurun_out/c18/exp.log
