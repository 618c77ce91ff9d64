#!/usr/bin/env python3
"""GPU fuzz beyond the suite's committed seeds: random rule sets through the HIP engine (C ABI) against the oracle, for a time budget.
usage (GPU box): python tools/gpufuzz.py <first seed> <seconds column grammar> <seconds mixed / residual grammar>
Three legs: the column compiler's grammar (tests/test_gpu_parity.py: test_fuzz_gpu_matches_oracle, other seeds), and rule sets mixing
column and residual rules (tests/test_gpu_residual.py: test_mixed_rule_sets_on_the_device) with the residual programs interpreted and
specialized by hiprtc. Prints one JSON line: seeds, requests and mismatches per leg (0 expected). The oracle is the checker."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import helpers as H  # noqa: E402
import test_residual as TR  # noqa: E402
from oracle import pyoracle  # noqa: E402
from pingoo_amd import RequestBatch, _abi  # noqa: E402
from pingoo_amd.engine import RuleEngine  # noqa: E402

first, t_col, t_mix = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
out = {}


def column_seed(seed):
    rng = random.Random(7_000_000 + seed)
    lists = H.fuzz_lists(rng)
    geo = H.fuzz_geoip(rng) if rng.random() < 0.7 else None
    with_geo = rng.random() < 0.3
    rules = [(f"r{k}", H.rexpr(rng, lists) if rng.random() < 0.95 else None, H.fuzz_actions(rng)) for k in range(rng.randint(1, 14))]
    flags = rng.choice([0, 0, _abi.OPT_NO_UA_GATE, _abi.OPT_NO_CAPTCHA_BYPASS, _abi.OPT_FILTER_STRIDE2])
    eng = RuleEngine(rules, lists, geo, flags=flags | _abi.OPT_LENIENT, lds_table_budget=rng.choice([0, 0, 1024, 2048]), max_table_bytes=rng.choice([0, 0, 4096]),
                     max_dfa_states=rng.choice([0, 0, 60]))
    batch = RequestBatch.from_requests(H.fuzz_requests(rng, rng.choice([1, 63, 64, 65, 200, 777]), with_geo))
    seen, _ = H.as_the_engine_sees(rules, eng.program)
    want = pyoracle.Oracle(seen, lists, geo, flags=flags & ~_abi.OPT_FILTER_STRIDE2).evaluate(batch)
    got, counts = eng.evaluate_batch(batch, with_counts=True)
    eng.close()
    return batch.n, int((got["action"] != want["action"]).sum() + (got["rule_idx"] != want["rule_idx"]).sum()) + int(counts.tolist() != np.bincount(want["action"], minlength=4).tolist())


def mixed_seed(seed, jit_flag):
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
    eng = RuleEngine(rules, TR.LISTS, geo, flags=flags | _abi.OPT_LENIENT | jit_flag)
    seen, _ = H.as_the_engine_sees(rules, eng.program)
    reqs = TR.requests(rng, rng.choice([1, 64, 65, 300, 1000]))
    if geo is not None:
        for r in reqs:
            r.asn = r.country = None
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(seen, TR.LISTS, geo, flags=flags).evaluate(batch)
    got = eng.evaluate_batch(batch)
    eng.close()
    return batch.n, int((got["action"] != want["action"]).sum() + (got["rule_idx"] != want["rule_idx"]).sum())


for name, fn, budget in (("column_grammar", column_seed, t_col), ("mixed_interpreted", lambda s: mixed_seed(s, _abi.OPT_NO_RESIDUAL_JIT), t_mix / 2),
                         ("mixed_specialized", lambda s: mixed_seed(s, 0), t_mix / 2)):
    t0, seeds, reqs, bad, bad_seeds = time.time(), 0, 0, 0, []
    while time.time() - t0 < budget:
        n, b = fn(first + seeds)
        if b:
            bad += b
            bad_seeds.append(first + seeds)
        seeds += 1
        reqs += n
    out[name] = {"first_seed": first, "seeds": seeds, "requests": reqs, "mismatches": bad, "mismatching_seeds": bad_seeds[:20], "seconds": round(time.time() - t0, 1)}
print(json.dumps(out))
