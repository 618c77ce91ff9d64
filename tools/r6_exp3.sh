# round-6 experiment 3 (GPU box): verdict2 sections alone; placements without events
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6_exp3; mkdir -p $O
A="--steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
export PWAF_LIB_VARIANT=prof
for skip in 0 1 2 8 16 32 59; do
  PWAF_PLACEMENT=1 PWAF_DEBUG_SKIP=$skip python bench.py $A > $O/skip_$skip.json 2> $O/skip_$skip.err
  python - <<PY
import json
d=json.load(open("$O/skip_$skip.json")); k=d["traffic_modes"]["tuned_benign"]["kernels_ms_per_step"]
print("skip $skip", round(d["ms_per_step"],4), k)
PY
done
for pl in 2 3 0; do
  PWAF_PLACEMENT=$pl PWAF_BENCH_NO_EVENTS=1 python bench.py $A > $O/pl_$pl.json 2> $O/pl_$pl.err
  python -c "
import json
d=json.load(open('$O/pl_$pl.json')); print('placement $pl noev', round(d['ms_per_step'],4))"
done
for ab in 256 512 768 1024; do
  PWAF_ATTR_BLOCKS=$ab PWAF_BENCH_NO_EVENTS=1 python bench.py $A > $O/ab_$ab.json 2> $O/ab_$ab.err
  python -c "
import json
d=json.load(open('$O/ab_$ab.json')); print('attr blocks $ab noev', round(d['ms_per_step'],4))"
done
