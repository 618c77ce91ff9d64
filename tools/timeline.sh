# usage (GPU box): bash tools/timeline.sh  — kernel-trace only; prints the last pipeline pass as a timeline (start offset, duration)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/timeline -o t --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie $BENCH_EXTRA > $R/gpurun_out/timeline.log 2>&1
python - <<PY
import csv, glob
rows = []
for p in glob.glob("$R/gpurun_out/timeline/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "verdict" in r["Kernel_Name"]]
lo, hi = idx[-(2 + int('$TL_BACK' or 0))] + 1, idx[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = None
for r in rows[lo:hi + 2]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f" gap {(s - prev_end) / 1e3:7.1f} us"
    print(f"q{r.get('Queue_Id','?'):<3} s{r.get('Stream_Id','?'):<3} {r['Kernel_Name'][:40]:<40} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us{gap}")
    if "attr" not in r["Kernel_Name"]:
        prev_end = e
PY
