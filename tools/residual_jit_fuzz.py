#!/usr/bin/env python3
"""Large CPU fuzz of the SPECIALIZED form of the residual rules (pingoo_amd/csrc/residual_jit.cpp): rule sets of the grammar of
tests/test_residual.py are translated, compiled with g++ (the harness of tests/test_residual_jit.py) and compared — three-way result
per rule and request — with the interpreter and with the oracle. usage: python tools/residual_jit_fuzz.py LO HI  (0 mismatches expected)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
import test_residual as T  # noqa: E402
import test_residual_jit as J  # noqa: E402
from pingoo_amd import RequestBatch  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
bad = n_sets = n_rules = 0
for seed in range(lo, hi):
    rng = random.Random(0x717E00 + seed)
    general = seed % 3 == 2  # every third set from the column compiler's own grammar (the interpreter is a complete evaluator)
    lists = H.fuzz_lists(rng) if general else T.LISTS
    exprs = J.accepted([(H.rexpr(rng, lists) if general else T.dbool(rng)) for _ in range(16)], lists)
    if not exprs:
        continue
    batch = RequestBatch.from_requests(H.fuzz_requests(rng, 32, with_geo=True) if general else T.requests(rng, 32))
    n_sets += 1
    n_rules += len(exprs)
    tag = f"fz{os.getpid()}_{seed}"  # (a fresh library per rule set: dlopen returns the already loaded image for a path it has seen)
    try:
        J.check_rule_set(exprs, lists, batch, tag)
    except AssertionError as exc:
        bad += 1
        print("MISMATCH seed", seed, str(exc)[:600], flush=True)
    finally:
        for f in (f"spec_{tag}.cpp", f"libspec_{tag}.so"):
            try:
                os.remove(os.path.join(J.BUILD, f))
            except OSError:
                pass
print("done", lo, hi, "rule sets", n_sets, "rules", n_rules, "mismatching sets", bad, "time", round(time.time() - t0, 1), flush=True)
