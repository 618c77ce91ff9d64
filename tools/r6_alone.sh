# round-6 (GPU box): every kernel alone (profiling build, PWAF_PLACEMENT=1)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/${TAG:-r6_alone}; mkdir -p $O
PWAF_LIB_VARIANT=prof PWAF_PLACEMENT=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $BENCH_ARGS > $O/alone.json 2> $O/alone.err
python -c "
import json
d=json.load(open('$O/alone.json')); print('alone', round(d['ms_per_step'],4), d['traffic_modes']['tuned_benign']['kernels_ms_per_step'])"
