# round-6 (GPU box): verdict / attr instruction counts, lazy vs eager comparison atoms
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6_pmc_ab; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export PWAF_LIB_VARIANT=prof PWAF_PLACEMENT=1
for fl in 0 32768; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc_$fl -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 --engine-flags $fl > $OUT/pmc_$fl.log 2>&1
  python - <<PY
import csv, glob, collections
d = collections.defaultdict(dict)
for f in glob.glob("$OUT/pmc_$fl/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "verdict" in r["Kernel_Name"] or "attr_kernel" in r["Kernel_Name"]: d[(int(r["Dispatch_Id"]), r["Kernel_Name"][:24])][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(d)[-2:]:
    v = d[k]; print("flags $fl", k[1], " ".join(f"{c.replace('SQ_','')}={v[c]:.3g}" for c in sorted(v)))
PY
  rm -rf $OUT/pmc_$fl
done
