# usage (GPU box): bash tools/prof_confirm.sh <tag> [bench args]  — kernel trace + issue counters of a short bench run
TAG=${1:-pc}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $@"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $DB "bench.py $ARGS" > $OUT/kernel_stats.txt 2>> $OUT/trace.log
head -30 $OUT/kernel_stats.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$name -o p --output-format csv -- python $R/bench.py $ARGS > $OUT/pmc_$name.log 2>&1
done
python $R/tools/pmc_report.py $OUT > $OUT/counters.txt 2>> $OUT/trace.log
grep -A14 "confirm_kernel" $OUT/counters.txt | head -60
rm -rf $OUT/trace
