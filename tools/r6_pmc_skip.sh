# round-6 (GPU box): instruction counts of the verdict kernel per section switch (profiling build)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6_pmc_skip; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export PWAF_LIB_VARIANT=prof PWAF_PLACEMENT=1
for skip in 0 1 2 8 16 32 59; do
  PWAF_DEBUG_SKIP=$skip rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc_$skip -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 > $OUT/pmc_$skip.log 2>&1
  python - <<PY
import csv, glob, collections
d = collections.defaultdict(dict)
for f in glob.glob("$OUT/pmc_$skip/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "verdict" in r["Kernel_Name"]: d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
k = sorted(d)[-1]; v = d[k]
print("skip $skip", " ".join(f"{c.replace('SQ_','')}={v[c]:.3g}" for c in sorted(v)))
PY
  rm -rf $OUT/pmc_$skip
done
