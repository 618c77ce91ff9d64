# usage: bash tools/pmc_scan.sh   (on the GPU box, from the repo root) — separate PMC passes, kernel-trace only (see task notes)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
SETS=${PMC_SETS:-"SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_WAIT_INST_LDS,SQ_INSTS_VALU,SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VALU,SQ_INSTS_SALU,SQ_INSTS_VMEM_RD,SQ_ACTIVE_INST_VMEM,SQ_WAVES GRBM_GUI_ACTIVE,GRBM_COUNT FETCH_SIZE WRITE_SIZE"}
for set in $SETS; do
  name=$(echo $set | cut -d, -f1)
  rocprofv3 --kernel-trace --pmc $(echo $set | tr ',' ' ') -d $R/gpurun_out/pmc_$name -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie > $R/gpurun_out/pmc_$name.log 2>&1
done
ls $R/gpurun_out/ | grep pmc_
