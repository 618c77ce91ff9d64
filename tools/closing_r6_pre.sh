# round-6 closing, in front of part A (GPU box): the resolve-heavy test files with one wave per slab forced, GPU fuzz both ways at new seeds
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6_final; mkdir -p $O
PWAF_RESOLVE_PARTS=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prefilter.py tests/test_gpu_paths.py -m gpu -x -q > $O/tests_parts1.log 2>&1; echo "rc=$?" >> $O/tests_parts1.log; grep -E "passed|failed|rc=" $O/tests_parts1.log
