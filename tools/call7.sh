cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/c7/pytest.log
cat gpurun_out/c7/pytest.log
export PWAF_ATTR_INLINE=1
bash tools/exp_round3.sh c7 base s2_015:PWAF_STRIDE2_FIELDS=0x15 s2_017:PWAF_STRIDE2_FIELDS=0x17 s2_014:PWAF_STRIDE2_FIELDS=0x14 s2_010:PWAF_STRIDE2_FIELDS=0x10 ipnov4:PWAF_DEBUG_SKIP=0x10000 ipnov6:PWAF_DEBUG_SKIP=0x20000 ipnone:PWAF_DEBUG_SKIP=0x30000 > gpurun_out/c7/exp.log 2>&1
cat gpurun_out/c7/exp.log
