"""CPU fuzz of the LOCALIZED list-scan walks: random rule sets, long fields, every alignment of a field in its arena.
The compiled tables are interpreted by tests/table_walker.py, which takes the shortest walk the device may take (it only sees the
field's own flagged chunks and checks the stop condition after every byte), and compared with the oracle.
usage: python tools/localfuzz.py <first seed> <last seed>"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from oracle import pyoracle
from pingoo_amd import Request, RequestBatch, _abi
from pingoo_amd.engine import CompiledProgram, UnsupportedExpression
from table_walker import Tables

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = skipped = 0
steps = [0, 0]
t0 = time.time()
for seed in range(lo, hi):
    rng = random.Random(seed)
    literal = seed % 3 == 0
    if literal:
        rules, lists = H.lit_rules(rng, rng.randint(2, 24)), {}
        alpha = "abcdefxyz/.=-_ %0123456789"
    else:
        lists = H.fuzz_lists(rng)
        rules = [(f"r{k}", H.rexpr(rng, lists), H.fuzz_actions(rng)) for k in range(rng.randint(1, 10))]
        alpha = H.ALPHA + ("xy" if rng.random() < 0.5 else "")
    flags = _abi.OPT_FILTER_STRIDE2 if rng.random() < 0.3 else 0
    try:
        prog = CompiledProgram(rules, lists, None, flags=flags | _abi.OPT_LENIENT, max_dfa_states=rng.choice([0, 0, 0, 60]))
    except UnsupportedExpression:
        skipped += 1
        continue

    def field(max_len):
        parts = []
        for _ in range(rng.randint(1, 5)):
            k = rng.random()
            if literal and k < 0.5:
                parts.append(H.lit_token(rng))
            elif literal and k < 0.6:
                parts.append(rng.choice(H.TOKENS).swapcase())
            parts.append(H.rstr(rng, 0, 70, alpha))
        return "".join(parts)[:max_len]

    def requests(n):
        out = []
        for _ in range(n):
            path = field(200)
            out.append(Request(host=field(60), url=path + ("?" + field(250) if rng.random() < 0.7 else ""), path=path, method=rng.choice(["GET", "POST"]),
                               user_agent=field(255), headers={"x-a": field(100)} if rng.random() < 0.4 else None))
        return out

    if rng.random() < 0.4:
        prog.tune(RequestBatch.from_requests(requests(200)))
    batch = RequestBatch.from_requests(requests(40))
    rules2, _ = H.as_the_engine_sees(rules, prog)
    want = pyoracle.Oracle(rules2, lists, None, flags=flags).evaluate(batch)
    t = Tables(prog)
    for local in (False, True):
        t.use_local_walks, t.n_steps = local, 0
        t.arena_offset = rng.randrange(16) if local else 0
        t.filter_phase = 0
        got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
        steps[local] += t.n_steps
        try:
            H.assert_verdicts_equal(got, want, batch, f"seed {seed} local {local} offset {t.arena_offset}")
        except AssertionError as ex:
            bad += 1
            print("MISMATCH seed", seed, "local", local, str(ex)[:400], flush=True)
print("done", lo, hi, "mismatches", bad, "unsupported", skipped, "steps whole/local", steps, "time", round(time.time() - t0, 1), flush=True)
