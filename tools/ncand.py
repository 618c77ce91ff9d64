import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["PWAF_DEBUG_SKIP"] = "64"
from synth import pysynth
from pingoo_amd.engine import RuleEngine
w = pysynth.Workload(3)
eng = RuleEngine(w.rules, w.lists, w.geoip)
b = w.batch(0, 64 * 2000)
v = eng.evaluate_batch(b)
nc = v["rule_idx"][::64].astype(np.int64)
print("candidates per group: mean %.1f median %d p90 %d max %d" % (nc.mean(), np.median(nc), np.percentile(nc, 90), nc.max()))
