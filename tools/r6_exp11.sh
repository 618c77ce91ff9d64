# round-6 diagnostic (GPU box, profiling build; WRONG verdicts with the skip switches): which side kernel stretches the first list scan
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6p; mkdir -p $O
A="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
run() { tag=$1; shift
  env "$@" python bench.py $A > $O/b_$tag.json 2> $O/b_$tag.err
  python -c "
import json
d=json.load(open('$O/b_$tag.json')); print('$tag', round(d['ms_per_step'],4), d['traffic_modes'][list(d['traffic_modes'])[0]]['kernels_ms_per_step'])"
}
run prof PWAF_LIB_VARIANT=prof
run skip_attr PWAF_LIB_VARIANT=prof PWAF_SKIP_ATTR=1
run skip_ipres PWAF_LIB_VARIANT=prof PWAF_SKIP_IPRES=1
run attr_prio PWAF_LIB_VARIANT=prof PWAF_ATTR_PRIO=1
run side_high PWAF_LIB_VARIANT=prof PWAF_SIDE_PRIORITY_HIGH=1
