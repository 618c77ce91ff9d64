# round-6 experiment 2 (GPU box): where the verdict / attribute kernels' time goes (profiling build: section switches, wrong results), everything alone
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6_exp2; mkdir -p $O
A="--steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
export PWAF_LIB_VARIANT=prof PWAF_PLACEMENT=1
for skip in 0 1 2 3 8 11 16 32 59 0x40000 0x80000 0x100000 0x1C0000 0x10000; do
  PWAF_DEBUG_SKIP=$skip python bench.py $A > $O/skip_$skip.json 2> $O/skip_$skip.err
  python - <<PY
import json
d=json.load(open("$O/skip_$skip.json")); k=d["traffic_modes"]["tuned_benign"]["kernels_ms_per_step"]
print("skip $skip", round(d["ms_per_step"],4), {x:k[x] for x in ("ipres","attr","verdict")})
PY
done
