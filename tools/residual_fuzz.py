#!/usr/bin/env python3
"""Large CPU fuzz of the residual interpreter (pingoo_amd/csrc/residual.h, host build of tests/rvm_host.cpp) against the oracle:
the grammar of tests/test_residual.py over many seeds. usage: python tools/residual_fuzz.py LO HI   -> prints mismatches (0 expected)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
import test_residual as T  # noqa: E402
from oracle import pyoracle  # noqa: E402
from pingoo_amd import RequestBatch, _abi  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
bad = n_expr = n_rej = n_eval = 0
for seed in range(lo, hi):
    rng = random.Random(0x5E5100 + seed)
    batch = RequestBatch.from_requests(T.requests(rng, 24))
    for _ in range(10):
        e = T.dbool(rng)
        try:
            pyoracle.compile_expression(e)
        except pyoracle.OracleError:
            continue
        n_expr += 1
        try:
            m = T.HostVM([e], T.LISTS)
        except ValueError:
            n_rej += 1
            continue
        orc = pyoracle.Oracle([("r", e, [H.B])], T.LISTS, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
        m.bind(batch)
        for i in range(batch.n):
            n_eval += 1
            if m.eval(0, i) != (orc.execute_rule(0, batch, i) == 1):
                bad += 1
                print("MISMATCH seed", seed, repr(e), i, flush=True)
                break
print("done", lo, hi, "expressions", n_expr, "rejected", n_rej, "evaluations", n_eval, "mismatches", bad, "time", round(time.time() - t0, 1), flush=True)
