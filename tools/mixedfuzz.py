#!/usr/bin/env python3
"""CPU fuzz of rule sets MIXING column rules and residual rules (the generator of tools/gpufuzz.py's mixed legs and of
tests/test_gpu_residual.py: test_mixed_rule_sets_on_the_device) through the compiled program — tables interpreted by
tests/table_walker.py, residual programs by the host build of residual.h — against the oracle.
usage: [PWAF_FUZZ_TUNE=1] python tools/mixedfuzz.py <first seed> <last seed>   (0 mismatches expected; found round 5's constant-folded header keys)"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import helpers as H  # noqa: E402
import table_walker  # noqa: E402
import test_residual as TR  # noqa: E402
from oracle import pyoracle  # noqa: E402
from pingoo_amd import RequestBatch, _abi  # noqa: E402
from pingoo_amd.engine import CompiledProgram  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0, bad, n_rules, n_refused = time.time(), [], 0, 0
for seed in range(lo, hi):
    rng = random.Random(8_000_000 + seed)
    rules = []
    for k in range(rng.randint(2, 12)):
        e = TR.dbool(rng) if rng.random() < 0.5 else H.rexpr(rng, TR.LISTS)
        try:
            pyoracle.compile_expression(e)
        except pyoracle.OracleError:
            e = "true"
        rules.append((f"r{k}", e, H.fuzz_actions(rng)))
    geo = H.fuzz_geoip(rng) if seed % 2 else None
    flags = rng.choice([0, _abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS])
    prog = CompiledProgram(rules, TR.LISTS, geo, flags=flags | _abi.OPT_LENIENT)
    try:
        seen, _ = H.as_the_engine_sees(rules, prog)
    except AssertionError:  # a rule beyond the interpreter's documented limits (64 heap items, 24 stack slots, 4 levels): refused, by index —
        n_refused += 1      # the set is skipped (the oracle would have to forget the refused rule's header names as well)
        continue
    reqs = TR.requests(rng, min(rng.choice([1, 64, 65, 300, 1000]), 65))  # (the walker is Python: bounded batches)
    if geo is not None:
        for r in reqs:
            r.asn = r.country = None
    batch = RequestBatch.from_requests(reqs)
    if os.environ.get("PWAF_FUZZ_TUNE"):  # profile-guided tables first (speed only, never a verdict)
        prog.tune(RequestBatch.from_requests(TR.requests(rng, rng.choice([1, 8, 64]))))
    orc = pyoracle.Oracle(seen, TR.LISTS, geo, flags=flags)
    want = orc.evaluate(batch)
    t = table_walker.Tables(prog)
    got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    n_rules += len(rules)
    if prog.header_names != orc.header_names or ((got["action"] != want["action"]) | (got["rule_idx"] != want["rule_idx"])).any():
        bad.append(seed)
        print("MISMATCH seed", seed, prog.header_names, orc.header_names, [r[1] for r in rules], flush=True)
print("done", lo, hi, "rules", n_rules, "rule sets with a refused rule", n_refused, "mismatching seeds", len(bad), bad[:10], "time", round(time.time() - t0, 1))
