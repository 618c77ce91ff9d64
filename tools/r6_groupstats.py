"""GPU box, profiling build: per 64-request group of the config-3 stream, the verdict kernel's candidate and entry counts
(PWAF_DEBUG_SKIP=64 writes n_cand | n_entries << 16 in place of the deciding rule)."""
import os, sys
os.environ["PWAF_LIB_VARIANT"] = "prof"
os.environ["PWAF_DEBUG_SKIP"] = "64"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from synth import pysynth
from pingoo_amd.engine import RuleEngine
w = pysynth.Workload(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
eng = RuleEngine(w.rules, w.lists, w.geoip)
eng.tune(w.batch(5_000_000, 32768))
for label, kw in (("benign", {}), ("hostile", {"adversarial": True})):
    b = w.batch(0, 640_000, **kw)
    got = eng.evaluate_batch(b)
    r = got["rule_idx"][::64].astype(np.int64)
    nc, ne = r & 0xFFFF, r >> 16
    print(label, "candidates/group mean %.1f p50 %d p90 %d p99 %d max %d | entries/group mean %.1f p50 %d p90 %d p99 %d max %d" % (
        nc.mean(), np.percentile(nc, 50), np.percentile(nc, 90), np.percentile(nc, 99), nc.max(), ne.mean(), np.percentile(ne, 50), np.percentile(ne, 90), np.percentile(ne, 99), ne.max()))
