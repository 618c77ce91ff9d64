# round-6 (GPU box): texture-addresser / L1 counters of the tail kernels (is the attribute kernel bound by its CU's vector-memory pipeline?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_ta; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TD_[A-Z0-9_]+)\b" | sort -u > $O/avail.txt; wc -l $O/avail.txt
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/pmc_$i -o p --output-format csv -- $B > $O/pmc_$i.log 2>&1
  python - <<PY
import csv, glob, collections
rows=[]
for p in glob.glob("$O/pmc_$i/**/*counter_collection.csv", recursive=True): rows += list(csv.DictReader(open(p)))
if not rows: print("set $i: no counters collected"); raise SystemExit
# last dispatch of each product kernel
last = {}
for r in rows:
    k = r["Kernel_Name"]
    if "pwaf::" not in k: continue
    last.setdefault(k.split("(")[0][:44], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, c in last.items():
    print(k.ljust(46), " ".join(f"{n}={v[-1]:.4g}" for n, v in c.items()))
PY
done 2>&1 | tee $O/summary.txt
rm -rf $O/pmc_*/
