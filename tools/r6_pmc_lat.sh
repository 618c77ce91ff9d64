# round-6 diagnostic (GPU box): average L1 -> L2 read latency per kernel (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ), in the step and alone (profiling build, one stream)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_lat; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
for mode in step alone; do
  if [ $mode = alone ]; then export PWAF_LIB_VARIANT=prof PWAF_PLACEMENT=1; fi
  timeout 150 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum -d $O/pmc_$mode -o p --output-format csv -- $B > $O/pmc_$mode.log 2>&1; echo "$mode rc=$?"
  python - <<PY
import csv, glob
rows=[]
for p in glob.glob("$O/pmc_$mode/**/*counter_collection.csv", recursive=True): rows += list(csv.DictReader(open(p)))
last = {}
for r in rows:
    k = r["Kernel_Name"]
    if "pwaf::" not in k: continue
    key = k.split("(")[0][:44] + ("#" + r["Dispatch_Id"] if "lscan_kernel" in k else "")
    last.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
# keep the last two lscan dispatches apart
for k, c in last.items():
    rq, lt = c.get("TCP_TCC_READ_REQ_sum", 0), c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0)
    if rq: print("$mode", k.ljust(52), f"read_req={rq:.4g} latency_sum={lt:.4g} avg_cycles={lt / rq:.0f}")
PY
done 2>&1 | tee $O/summary.txt
rm -rf $O/pmc_*/
