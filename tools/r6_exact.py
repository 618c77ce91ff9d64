import os, sys
os.environ["PWAF_LIB_VARIANT"] = "prof"; os.environ["PWAF_DEBUG_SKIP"] = "512"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from synth import pysynth
from pingoo_amd.engine import RuleEngine
w = pysynth.Workload(3)
eng = RuleEngine(w.rules, w.lists, w.geoip); eng.tune(w.batch(5_000_000, 32768))
b = w.batch(0, 640_000); got = eng.evaluate_batch(b)
r = got["rule_idx"][::64].astype(np.int64)
print("exact passes/group mean %.2f p50 %d p90 %d max %d" % (r.mean(), np.percentile(r,50), np.percentile(r,90), r.max()))
