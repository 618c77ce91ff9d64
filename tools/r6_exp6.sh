# round-6 experiment (GPU box): compact list-scan items; side-stream occupancy (profiling build: PWAF_IPRES_BLOCKS / PWAF_ATTR_BLOCKS)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6k; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log
A="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
run() { # tag, env...
  tag=$1; shift
  env "$@" python bench.py $A $EXTRA > $O/b_$tag.json 2> $O/b_$tag.err
  python -c "
import json
d=json.load(open('$O/b_$tag.json')); print('$tag', round(d['ms_per_step'],4), d['traffic_modes'][list(d['traffic_modes'])[0]]['kernels_ms_per_step'])"
}
run product X=1
EXTRA="--requests 1250000" run product_1250000 X=1
run prof PWAF_LIB_VARIANT=prof
run ip1536 PWAF_LIB_VARIANT=prof PWAF_IPRES_BLOCKS=1536
run ip1024 PWAF_LIB_VARIANT=prof PWAF_IPRES_BLOCKS=1024
run ip1024_at1024 PWAF_LIB_VARIANT=prof PWAF_IPRES_BLOCKS=1024 PWAF_ATTR_BLOCKS=1024
run ip1536_at1024 PWAF_LIB_VARIANT=prof PWAF_IPRES_BLOCKS=1536 PWAF_ATTR_BLOCKS=1024
run at1024 PWAF_LIB_VARIANT=prof PWAF_ATTR_BLOCKS=1024
EXTRA="--adversarial" run product_adv X=1
BENCH_EXTRA="--no-config5" bash tools/timeline.sh > $O/timeline_10M.txt 2>&1; tail -15 $O/timeline_10M.txt
