cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c11 gpurun_out/c11adv
./tools/_bin/dpp_probe > gpurun_out/dpp_probe.txt 2>&1; cat gpurun_out/dpp_probe.txt
# stride-2 decomposition on the three long columns (host, path, user agent = fields 0, 2, 4 -> mask 0x15) + url (0x17)
bash tools/exp_round3.sh c11 base s2all:PWAF_STRIDE2_FIELDS=0x1f s2nolook:PWAF_STRIDE2_FIELDS=0x1f,PWAF_FILTER_DEBUG_SKIP=1 s2noload:PWAF_STRIDE2_FIELDS=0x1f,PWAF_FILTER_DEBUG_SKIP=2 s2noheads:PWAF_STRIDE2_FIELDS=0x1f,PWAF_FILTER_DEBUG_SKIP=4 s2none:PWAF_STRIDE2_FIELDS=0x1f,PWAF_FILTER_DEBUG_SKIP=7 vw6:PWAF_VERDICT_WAVES=6 vw9:PWAF_VERDICT_WAVES=9 > gpurun_out/c11/exp.log 2>&1
cat gpurun_out/c11/exp.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c11adv base advtune:PWAF_BENCH_TUNE_ADVERSARIAL=1 > gpurun_out/c11adv/exp.log 2>&1
cat gpurun_out/c11adv/exp.log
grep -h "pass\|filter\|candidate" gpurun_out/c11adv/advtune.err | head -40
