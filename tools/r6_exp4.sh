R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6_exp4; mkdir -p $O
A="--steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
export PWAF_LIB_VARIANT=prof PWAF_PLACEMENT=1
for cfg in "0 0" "32768 0" "0 1024" "0 2048" "0 3072"; do
  set -- $cfg
  PWAF_DEBUG_SKIP=$2 python bench.py $A --engine-flags $1 > $O/x.json 2> $O/x.err
  python -c "
import json
d=json.load(open('$O/x.json')); k=d['traffic_modes']['tuned_benign']['kernels_ms_per_step']; print('flags $1 skip $2', round(d['ms_per_step'],4), {x:k[x] for x in ('attr','verdict')})"
done
