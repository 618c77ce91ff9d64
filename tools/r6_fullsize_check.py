#!/usr/bin/env python3
"""Full-size property check of the headline batch (GPU box): the 10M-request benign batch of BASELINE configs[2] through the tuned engine
(bigram prefilter -> resolve -> confirm tier -> list scans) and through an engine WITHOUT prefilters (every pass walks every request through its
full DFA): every one of the 10M verdicts must agree — with resolve_kernel's launch shape left to the engine, forced to one wave per slab and forced
to four. The oracle checks a 20k random sample of the same batch in tests/test_gpu_prefilter.py and a prefix in the bench line; it cannot check 10M.
usage: python tools/r6_fullsize_check.py [n] [config id] [hostile]   -> one JSON line"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pingoo_amd import _abi  # noqa: E402
from pingoo_amd.engine import RuleEngine  # noqa: E402
from synth import pysynth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
hostile = len(sys.argv) > 3 and sys.argv[3] == "hostile"
w = pysynth.Workload(cfg)
batch = w.batch(0, n, adversarial=hostile)
eng = RuleEngine(w.rules, w.lists, w.geoip)
eng.tune(w.batch(n, 32768))
plain = RuleEngine(w.rules, w.lists, w.geoip, flags=_abi.OPT_NO_PREFILTER)
t0 = time.time()
ref = plain.evaluate_batch(batch)
out = {"config": cfg, "stream": "hostile" if hostile else "benign", "requests": n, "plain_seconds": round(time.time() - t0, 2), "non_allow": int((ref["action"] != 0).sum()), "legs": {}}
for tag, val in (("engine's choice", None), ("one wave per slab", "1"), ("four waves per slab", "4")):
    if val is None:
        os.environ.pop("PWAF_RESOLVE_PARTS", None)
    else:
        os.environ["PWAF_RESOLVE_PARTS"] = val
    got = eng.evaluate_batch(batch)
    bad = int(((got["action"] != ref["action"]) | (got["rule_idx"] != ref["rule_idx"])).sum())
    out["legs"][tag] = {"differing_verdicts": bad}
print(json.dumps(out))
eng.close()
plain.close()
