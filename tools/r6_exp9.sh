# round-6 experiment (GPU box, profiling build): LDS share of the first list-scan launch (R-tier walks: cold rows = L2 round trips in the walk's chain)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6n; mkdir -p $O
A="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
run() { tag=$1; shift
  env "$@" python bench.py $A $EXTRA > $O/b_$tag.json 2> $O/b_$tag.err
  python -c "
import json
d=json.load(open('$O/b_$tag.json')); print('$tag', round(d['ms_per_step'],4), d['traffic_modes'][list(d['traffic_modes'])[0]]['kernels_ms_per_step'])"
}
run prof PWAF_LIB_VARIANT=prof
run shape1 PWAF_LIB_VARIANT=prof PWAF_LIST_SHAPE=1
run shape17 PWAF_LIB_VARIANT=prof PWAF_LIST_SHAPE=17
run shape2 PWAF_LIB_VARIANT=prof PWAF_LIST_SHAPE=2
run shape3 PWAF_LIB_VARIANT=prof PWAF_LIST_SHAPE=3
run async2 PWAF_LIB_VARIANT=prof PWAF_LSCAN_ASYNC=2
EXTRA="--config 5" run c5_prof PWAF_LIB_VARIANT=prof
EXTRA="--config 5" run c5_shape1 PWAF_LIB_VARIANT=prof PWAF_LIST_SHAPE=1
