"""Extended CPU fuzz: compiled tables (interpreted by tests/table_walker.py) against the oracle on random rule sets.
usage: [PWAF_FUZZ_STRIDE2=1] [PWAF_FUZZ_TUNE=1] [PWAF_FUZZ_MANY=1] python tools/bigfuzz.py <first seed> <last seed>   (about 70 seeds per second and core; no GPU involved)"""
import sys, random, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import helpers as H
from oracle import pyoracle
from pingoo_amd import RequestBatch, _abi
from pingoo_amd.engine import CompiledProgram, UnsupportedExpression
from table_walker import Tables
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0=time.time()
for seed in range(lo, hi):
    rng = random.Random(seed)
    lists = H.fuzz_lists(rng)
    geo = H.fuzz_geoip(rng) if rng.random() < 0.7 else None
    with_geo = rng.random() < 0.3
    rules = []
    many = bool(os.environ.get("PWAF_FUZZ_MANY"))  # rule sets of 40 - 300 rules: pass splitting, atom de-duplication, rule bitmaps beyond one word
    for k in range(rng.choice([40, 80, 150, 300]) if many else rng.randint(1, 12)):
        e = H.rexpr(rng, lists) if rng.random() < 0.95 else None
        acts = H.fuzz_actions(rng)
        rules.append((f"r{k}", e, acts))
    flags = rng.choice([0, 0, _abi.OPT_NO_UA_GATE, _abi.OPT_NO_CAPTCHA_BYPASS, _abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS])
    if os.environ.get("PWAF_FUZZ_STRIDE2"):
        flags |= _abi.OPT_FILTER_STRIDE2  # stride-2 prefilters wherever they can be built; the walker samples from the seed's parity
    try:
        prog = CompiledProgram(rules, lists, geo, flags=flags | _abi.OPT_LENIENT, max_table_bytes=rng.choice([0, 0, 2048, 4096]), max_dfa_states=rng.choice([0, 0, 40]))
    except UnsupportedExpression:
        continue
    if os.environ.get("PWAF_FUZZ_TUNE"):
        # profile-guided tables (pwaf_program_tune: prefilter bigram statistics and heads, stride choice, LDS-resident rows) from a traffic
        # sample of random size: tuning may change speed, never a verdict
        prog.tune(RequestBatch.from_requests(H.fuzz_requests(rng, rng.choice([1, 5, 48, 200]), with_geo)))
    batch = RequestBatch.from_requests(H.fuzz_requests(rng, 48, with_geo))
    rules, _ = H.as_the_engine_sees(rules, prog)
    want = pyoracle.Oracle(rules, lists, geo, flags=flags).evaluate(batch)
    t = Tables(prog)
    t.filter_phase = seed & 1
    t.arena_offset, t.arena_chunks = (seed >> 1) % 16, (seed >> 5) % 3  # (where the field lies in its arena: the windows of short factors reach into the neighbouring bytes)
    got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    try:
        H.assert_verdicts_equal(got, want, batch, f"seed {seed}")
    except AssertionError as ex:
        bad += 1
        print("MISMATCH seed", seed, str(ex)[:300], flush=True)
print("done", lo, hi, "mismatches", bad, "time", round(time.time()-t0,1), flush=True)
