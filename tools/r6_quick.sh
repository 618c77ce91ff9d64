# round-6 quick check (GPU box): [tests] + short bench of config 3 (driver step counts, no side legs)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/${TAG:-r6_quick}; mkdir -p $O
if [ -n "$TESTS" ]; then python -m pytest tests -m gpu -x -q $TESTS_ARGS > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log; tail -5 $O/gputests.log; fi
A="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
python bench.py $A $BENCH_ARGS > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json")); print("bench", round(d["ms_per_step"],4), d["traffic_modes"]["tuned_benign"]["kernels_ms_per_step"], d["config"]["action_counts_allow_block_captcha_bypass"])
PY
