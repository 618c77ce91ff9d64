cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c12 gpurun_out/c12adv
( timeout 900 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/pytest12.log; cat gpurun_out/pytest12.log
bash tools/exp_round3.sh c12 rows segs:PWAF_FILTER_SEGMENTS=1 rnolook:PWAF_FILTER_DEBUG_SKIP=1 rnoload:PWAF_FILTER_DEBUG_SKIP=2 rnoheads:PWAF_FILTER_DEBUG_SKIP=4 rnone:PWAF_FILTER_DEBUG_SKIP=7 s2rows:PWAF_STRIDE2_FIELDS=0x1f s2rnoheads:PWAF_STRIDE2_FIELDS=0x1f,PWAF_FILTER_DEBUG_SKIP=4 s2rnone:PWAF_STRIDE2_FIELDS=0x1f,PWAF_FILTER_DEBUG_SKIP=7 > gpurun_out/c12/exp.log 2>&1
cat gpurun_out/c12/exp.log
