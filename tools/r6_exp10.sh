# round-6 experiment (GPU box): chunk-driven resolve with sampled offsets — suite, resolve-heavy files and fuzz with one wave per slab forced, step times (10M; 1.25M both ways)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6o; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log
PWAF_RESOLVE_PARTS=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prefilter.py tests/test_gpu_paths.py -m gpu -x -q > $O/tests_parts1.log 2>&1; echo "rc=$?" >> $O/tests_parts1.log; grep -E "passed|failed|rc=" $O/tests_parts1.log
PWAF_RESOLVE_PARTS=1 python tools/gpufuzz.py 730000 60 0 > $O/gpufuzz_parts1.json 2> $O/gpufuzz_parts1.err; cut -c1-200 $O/gpufuzz_parts1.json
A="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
run() { tag=$1; shift
  env "$@" python bench.py $A $EXTRA > $O/b_$tag.json 2> $O/b_$tag.err
  python -c "
import json
d=json.load(open('$O/b_$tag.json')); print('$tag', round(d['ms_per_step'],4), d['traffic_modes'][list(d['traffic_modes'])[0]]['kernels_ms_per_step'])"
}
run 10M X=1
EXTRA="--requests 1250000" run 1250000_default X=1
EXTRA="--requests 1250000" run 1250000_parts1 PWAF_RESOLVE_PARTS=1
EXTRA="--requests 2500000" run 2500000_default X=1
EXTRA="--requests 2500000" run 2500000_parts4 PWAF_RESOLVE_PARTS=4
TAG=r6o bash tools/r6_alone.sh
