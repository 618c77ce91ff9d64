# round-6 (GPU box): issue / LDS counters per product kernel for the current build: bash tools/r6_pmc.sh <tag> [bench args]
TAG=${1:-r6_pmc}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$name -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $@ > $OUT/pmc_$name.log 2>&1
done
python $R/tools/pmc_report.py $OUT > $OUT/counters.txt 2>> $OUT/pmc.log
cat $OUT/counters.txt
rm -rf $OUT/pmc_*/ 2>/dev/null
