# round-6 experiment 1 (GPU box): what the HIP events between kernels cost, alone times, timeline without events
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6_exp1; mkdir -p $O
A="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 --verbose"
python bench.py $A > $O/base.json 2> $O/base.err
PWAF_BENCH_NO_EVENTS=1 python bench.py $A > $O/noev.json 2> $O/noev.err
PWAF_LIB_VARIANT=prof PWAF_PLACEMENT=1 python bench.py $A > $O/alone.json 2> $O/alone.err
PWAF_LIB_VARIANT=prof PWAF_PLACEMENT=3 PWAF_BENCH_NO_EVENTS=1 python bench.py $A > $O/p3_noev.json 2> $O/p3_noev.err
PWAF_BENCH_NO_EVENTS=1 BENCH_EXTRA="--no-config5 --no-extra-modes" bash tools/timeline.sh > $O/timeline_noev.txt 2>&1
for f in base noev alone p3_noev; do python - <<PY
import json
d=json.load(open("$O/$f.json")); print("$f", round(d["ms_per_step"],4), d["traffic_modes"]["tuned_benign"]["kernels_ms_per_step"])
PY
done
cat $O/timeline_noev.txt | tail -25
