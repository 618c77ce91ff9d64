# usage (on the GPU box, from the repo root):  [PROFILE_ARGS="--config 5"] bash tools/profile_round.sh <tag>
# Produces, under gpurun_out/<tag>/: the official bench line, the rocprofv3 --kernel-trace --stats summary of the same command
# and the HBM-traffic counters (separate --pmc passes, kernel-trace only — never combined with other trace domains).
TAG=${1:-r1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py $PROFILE_ARGS > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $PROFILE_ARGS > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $DB "bench.py --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $PROFILE_ARGS (default config 3, 10M requests x 1024 rules, steps 5 warmup 2)" > $OUT/kernel_stats.txt 2>> $OUT/trace.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $PROFILE_ARGS > $OUT/pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py $OUT > $OUT/traffic.json 2>> $OUT/trace.log
# issue / LDS counters of the same command (separate passes; SQ_* count quad-cycles, see MI355X_MICROARCH.md)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$name -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $PROFILE_ARGS > $OUT/pmc_$name.log 2>&1
done
python $R/tools/pmc_report.py $OUT > $OUT/counters.txt 2>> $OUT/trace.log
cat $OUT/kernel_stats.txt | head -30; cat $OUT/traffic.json
rm -rf $OUT/trace  # the database is large; the summary is what gets committed
