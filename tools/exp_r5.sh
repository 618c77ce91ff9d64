# round-5 experiment driver (GPU box): bash tools/exp_r5.sh <tag> <what...>   — scratch tool, results under gpurun_out/<tag>/
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
for what in "$@"; do
  case $what in
    tests_attr) python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_paths.py -m gpu -x -q > $O/pytest_attr.log 2>&1; echo "rc=$?" >> $O/pytest_attr.log; tail -3 $O/pytest_attr.log ;;
    tests) python -m pytest $R/tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log ;;
    timeline) bash $R/tools/timeline.sh > $O/timeline.txt 2>&1; tail -30 $O/timeline.txt ;;
    skip=*) v=${what#skip=}; PWAF_LIB_VARIANT=prof PWAF_DEBUG_SKIP=$v $B > $O/skip_$v.json 2> $O/skip_$v.err; PWAF_LIB_VARIANT=prof PWAF_PLACEMENT=1 PWAF_DEBUG_SKIP=$v $B > $O/alone_skip_$v.json 2> $O/alone_skip_$v.err ;;
    bench) python $R/bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json ;;
    flags=*) v=${what#flags=}; python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-config5 --residual 0 --engine-flags $v > $O/flags_$v.json 2> $O/flags_$v.err ;;
    place=*) v=${what#place=}; PWAF_LIB_VARIANT=prof PWAF_PLACEMENT=$v $B > $O/place_$v.json 2> $O/place_$v.err ;;
    quick) $B > $O/quick.json 2> $O/quick.err ;;
    profile=*) bash $R/tools/profile_round.sh $TAG/${what#profile=} ;;
  esac
done
