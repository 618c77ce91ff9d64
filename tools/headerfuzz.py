#!/usr/bin/env python3
"""CPU fuzz of HEADER-heavy rule sets (the extension of BASELINE configs[4]: up to 120 header names, up to 120 rules of predicates over
`http_request.headers[..]` — literals, regexes, lengths, membership, header against field) through the compiled program (tables by
tests/table_walker.py, overflow rules by the residual host VM) against the oracle.
usage: python tools/headerfuzz.py <first seed> <last seed>   (0 mismatches expected; found the field-against-field table overflow that
failed engine creation: round 5)"""
import sys, random, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import helpers as H
from oracle import pyoracle
from pingoo_amd import RequestBatch, Request, _abi
from pingoo_amd.engine import CompiledProgram, UnsupportedExpression
from table_walker import Tables
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0; t0=time.time(); skipped=0
for seed in range(lo, hi):
    rng = random.Random(11_000_000+seed)
    n_names = rng.choice([1, 5, 20, 64, 90, 120])
    names = [f"x-h{k}" for k in range(n_names)]
    def hpred():
        nm = rng.choice(names)
        f = rng.choice([f'http_request.headers["{nm}"]', f'http_request.headers["{nm}"]', f'http_request["headers"]["{nm}"]'])
        k = rng.randint(0, 9)
        if k <= 3: return f'{f}.{rng.choice(["contains","starts_with","ends_with"])}({H.q(H.rstr(rng,0,3))})'
        if k == 4: return f'{f} {rng.choice(["==","!="])} {H.q(H.rstr(rng,0,3))}'
        if k == 5: return f'{f}.matches({H.q(H.rregex(rng))})'
        if k == 6: return f'{f}.length() {rng.choice(["<",">","==","<=",">="])} {rng.randint(0,6)}'
        if k == 7: return f'"{nm}" in http_request.headers'
        if k == 8: return f'{f} == http_request.{rng.choice(["host","path","method"])}'
        return H.rpred(rng, None) if False else f'{f}.contains("a")'
    rules = []
    for k in range(rng.choice([3, 10, 40, 120])):
        e = hpred()
        for _ in range(rng.randint(0, 2)):
            e = f'({e} {rng.choice(["&&","||"])} {hpred()})'
        if rng.random() < 0.2: e = "!(" + e + ")"
        try: pyoracle.compile_expression(e)
        except pyoracle.OracleError: e = "true"
        rules.append((f"r{k}", e, H.fuzz_actions(rng)))
    flags = rng.choice([0, _abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS, _abi.OPT_FILTER_STRIDE2])
    try:
        prog = CompiledProgram(rules, None, None, flags=flags | _abi.OPT_LENIENT)
    except UnsupportedExpression as e:
        skipped += 1; print('SKIP create', seed, str(e)[:200], flush=True); continue
    reqs = []
    for _ in range(24):
        hdrs = {nm: H.rstr(rng, 0, 6, H.UALPHA if rng.random() < 0.3 else H.ALPHA) for nm in names if rng.random() < 0.5}
        reqs.append(Request(host=H.rstr(rng,0,4), path="/"+H.rstr(rng,0,4), url="/"+H.rstr(rng,0,6), method=rng.choice(["GET","POST"]), user_agent="ua", headers=hdrs or None))
    batch = RequestBatch.from_requests(reqs)
    if rng.random() < 0.5: prog.tune(batch)
    try: rules2, _ = H.as_the_engine_sees(rules, prog)
    except AssertionError as e: skipped += 1; print('SKIP refused', seed, str(e)[:300], flush=True); continue
    orc = pyoracle.Oracle(rules2, None, None, flags=flags & ~_abi.OPT_FILTER_STRIDE2)
    want = orc.evaluate(batch)
    t = Tables(prog)
    t.filter_phase = seed & 1
    got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    if prog.header_names != orc.header_names or ((got["action"] != want["action"]) | (got["rule_idx"] != want["rule_idx"])).any():
        bad += 1; print("MISMATCH seed", seed, len(names), len(rules), prog.header_names[:5], orc.header_names[:5], flush=True)
print("done", lo, hi, "skipped", skipped, "mismatches", bad, "time", round(time.time()-t0,1))
