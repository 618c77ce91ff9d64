#!/usr/bin/env python3
"""CPU fuzz of UNICODE regex rules through the compiled program (scalar-mode tables, bigram filter + confirm tier by
tests/table_walker.py) against the oracle: random patterns of tests/test_oracle.py's Unicode generator (classes beyond ASCII, (?i) folds
that reach ASCII, \\b next to non-ASCII, `.` over multi-byte scalars) on url / path / host / User-Agent values drawn from an alphabet
of 1- to 4-byte scalars; half of the programs tuned first. usage: python tools/unicodefuzz.py <first seed> <last seed>"""
import sys, random, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import helpers as H, test_oracle as T
from oracle import pyoracle
from pingoo_amd import RequestBatch, Request, _abi
from pingoo_amd.engine import CompiledProgram, UnsupportedExpression
from table_walker import Tables
lo, hi = int(sys.argv[1]), int(sys.argv[2])
alphabet = ["a", "s", "S", "k", "K", "ſ", "K", "é", "É", "σ", "ς", "Σ", "€", "٣", " ", " ", " ", ".", "_", "1", "\n", "\U0001F600", "/", "b"]
bad = 0; t0 = time.time(); skipped = 0; nrules = 0
for seed in range(lo, hi):
    rng = random.Random(77_000_000 + seed)
    rules = []
    for k in range(rng.randint(1, 10)):
        pat = T._rand_uregex(rng)
        f = rng.choice(["url", "path", "host", "user_agent"])
        e = f'http_request.{f}.matches({H.q(pat)})'
        if rng.random() < 0.3: e += f' && http_request.{rng.choice(["url","path"])}.contains({H.q("".join(rng.choice(alphabet) for _ in range(rng.randint(1,3))))})'
        try: pyoracle.compile_expression(e)
        except pyoracle.OracleError: continue
        rules.append((f"r{k}", e, H.fuzz_actions(rng)))
    if not rules: continue
    flags = rng.choice([0, _abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS, _abi.OPT_FILTER_STRIDE2])
    try: prog = CompiledProgram(rules, None, None, flags=flags | _abi.OPT_LENIENT, max_dfa_states=rng.choice([0, 0, 60]))
    except UnsupportedExpression as e: skipped += 1; continue
    def s(): return "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 12)))
    reqs = [Request(host=s(), url="/" + s(), path="/" + s(), method="GET", user_agent=s() or "u") for _ in range(32)]
    batch = RequestBatch.from_requests(reqs)
    if rng.random() < 0.5: prog.tune(batch)
    try: rules2, _ = H.as_the_engine_sees(rules, prog, allow=len(rules))
    except AssertionError: skipped += 1; continue
    want = pyoracle.Oracle(rules2, None, None, flags=flags & ~_abi.OPT_FILTER_STRIDE2).evaluate(batch)
    t = Tables(prog); t.filter_phase = seed & 1; t.arena_offset = (seed >> 1) % 16
    got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    nrules += len(rules)
    if ((got["action"] != want["action"]) | (got["rule_idx"] != want["rule_idx"])).any():
        bad += 1; print("MISMATCH seed", seed, [r[1] for r in rules][:3], flush=True)
print("done", lo, hi, "rules", nrules, "skipped", skipped, "mismatches", bad, "time", round(time.time()-t0,1))
