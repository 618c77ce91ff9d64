# the round's closing measurements: full GPU suite, the driver's bench command, rocprofv3 kernel stats + PMC traffic + issue counters of
# config 3 (benign), kernel stats + traffic of the same with 8 residual rules (rvm_jit_kernel)
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
tail -4 $O/gputests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/final/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"])
for k in ("residual","batcher","pcie_inclusive"):
    print(k, json.dumps(d.get(k))[:700])
print("config5", json.dumps(d.get("config5"))[:900])
print("adv", json.dumps(d["traffic_modes"]["adversarial_tuned_on_benign"])[:500])
P
bash tools/profile_round4.sh final_c3 > $O/profile_c3.log 2>&1; tail -26 $O/profile_c3.log
PROFILE_LIGHT=1 bash tools/profile_round4.sh final_c3_res8 --residual 8 > $O/profile_res8.log 2>&1; tail -26 $O/profile_res8.log
