# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh <tag> [bench args]      (PROFILE_LIGHT=1: kernel stats + HBM traffic only)
# Before the gpurun call:  git log -1 --format=%h -- pingoo_amd/csrc > .commit_id   (the box has no .git; tools/collect_profiles.py checks it)
# Produces under gpurun_out/<tag>/: the rocprofv3 --kernel-trace --stats summary of a short bench run (kernel_stats.txt), the HBM-traffic
# counters of the same command (traffic.json: separate --pmc passes, kernel-trace only — never combined with other trace domains) and
# the issue / LDS counters per product kernel (counters.txt). The bench line itself comes from a plain `python bench.py` run.
TAG=${1:-r6}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $@"   # (the driver's step counts: a 4-launch trace caught one box before its clocks settled — 0.624 ms for a kernel the 25-launch trace and the bench's own events put at 0.572 / 0.577)
export PWAF_COMMIT=$(cat $R/.commit_id 2>/dev/null || echo "?")
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $DB "bench.py $ARGS" > $OUT/kernel_stats.txt 2>> $OUT/trace.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $@ > $OUT/pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py $OUT > $OUT/traffic.json 2>> $OUT/trace.log
if [ -n "$PROFILE_LIGHT" ]; then head -24 $OUT/kernel_stats.txt; rm -rf $OUT/trace $OUT/pmc_*/ 2>/dev/null; exit 0; fi  # (kernel stats + HBM traffic only)
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$name -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $@ > $OUT/pmc_$name.log 2>&1
done
python $R/tools/pmc_report.py $OUT > $OUT/counters.txt 2>> $OUT/trace.log
head -24 $OUT/kernel_stats.txt
rm -rf $OUT/trace $OUT/pmc_*/ 2>/dev/null
