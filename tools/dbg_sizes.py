import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pingoo_amd.engine import DeviceBatch, RuleEngine
from synth import pysynth
wl = pysynth.Workload(3)
eng = RuleEngine(wl.rules, wl.lists, wl.geoip)
b = wl.batch(0, 2_000_000)
db = DeviceBatch(b, "cuda:0")
out = eng.evaluate_device(db)
torch.cuda.synchronize()
print("device ok", flush=True)
for m in (1000, 100_000, 1_000_000):
    hb = b.slice(0, m)
    print("host", m, flush=True)
    v = eng.evaluate_batch(hb)
    print("ok", m, int(v["action"].sum()), flush=True)
eng.tune(wl.batch(5_000_000, 8192))
print("tuned", flush=True)
v = eng.evaluate_batch(b.slice(0, 1_000_000))
print("ok tuned", flush=True)
