"""The residual rules on the device (through the C ABI): rules the column compiler cannot take are lowered to stack programs
(residual.h) and evaluated per request — by default as the SPECIALIZED program hiprtc compiled for the device when the engine was
created (residual_jit.cpp; pwaf_engine_residual_mode == 2), with PWAF_OPT_NO_RESIDUAL_JIT by the interpreter kernel (mode 1) — and
land in a pseudo pass of the verdict kernel. Verdicts must be the oracle's — whichever path evaluates a rule."""
import random

import numpy as np
import pytest

import helpers as H
import test_residual as TR
from oracle import pyoracle
from pingoo_amd import Request, RequestBatch, _abi
from pingoo_amd.engine import RuleEngine

pytestmark = pytest.mark.gpu
B, CAP = _abi.RULE_ACTION_BLOCK, _abi.RULE_ACTION_CAPTCHA


JIT = {"specialized": 0, "interpreted": _abi.OPT_NO_RESIDUAL_JIT}


def check_mode(eng, how):
    n_res = sum("residual interpreter" in w for w in eng.program.warnings())
    assert eng.residual_mode == (0 if n_res == 0 else 2 if how == "specialized" else 1), (how, eng.residual_mode, eng.program.warnings())


@pytest.mark.parametrize("seed,how", [(s, "specialized") for s in range(16)] + [(s, "interpreted") for s in range(0, 16, 3)])
def test_mixed_rule_sets_on_the_device(seed, how):
    rng = random.Random(717100 + seed)
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
    eng = RuleEngine(rules, TR.LISTS, geo, flags=flags | _abi.OPT_LENIENT | JIT[how])
    check_mode(eng, how)
    seen, _ = H.as_the_engine_sees(rules, eng.program)
    n = rng.choice([1, 64, 65, 300, 1000])
    reqs = TR.requests(rng, n)
    if geo is not None:  # the engine resolves client.asn / client.country itself: through the trie with RECORD leaves
        for r in reqs:
            r.asn = r.country = None
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(seen, TR.LISTS, geo, flags=flags).evaluate(batch)
    got, counts = eng.evaluate_batch(batch, with_counts=True)
    H.assert_verdicts_equal(got, want, batch, f"seed {seed}: {[r[1] for r in rules]}")
    assert counts.tolist() == np.bincount(want["action"], minlength=4).tolist()
    eng.close()


@pytest.mark.parametrize("how", list(JIT))
def test_residual_known_answers_and_dnf_explosion_on_the_device(how):
    rules = [("arith", "http_request.path.length() + 1 > http_request.url.length() && client.remote_port % 2 == 0", [B]),
             ("concat", '(http_request.host + ":" + http_request.method).matches("^[a-z]+:(GET|POST)$") && http_request.path + "x" == "/qx"', [CAP]),
             ("list", '[http_request.host, "zz"].contains(http_request.path)', [B]),
             ("order", 'http_request.host < http_request.path && http_request.path < "c"', [CAP]),
             ("country", "http_request.url.contains(client.country) && client.asn * 2 == 128", [B]),
             ("cond", '(http_request.path.starts_with("/a") ? http_request.host : http_request.url).ends_with("!")', [B]),
             ("big", " && ".join(f'(http_request.path.contains("a{k}") || http_request.url.contains("b{k}") || http_request.host.contains("c{k}"))' for k in range(8)), [CAP]),
             ("plain", 'http_request.path.contains("plain")', [B])]
    eng = RuleEngine(rules, flags=JIT[how])
    check_mode(eng, how)
    assert not eng.partial and sum("residual interpreter" in w for w in eng.program.warnings()) == 7
    rng = random.Random(5)
    words = ["/q", "a", "b", "zz", "ab:", "x!", "/a!", "FR", "plain", "a0a1a2a3a4a5a6a7", "b0", "c1c2", "/abc", ""]
    reqs = [Request(host=rng.choice(words) + rng.choice(words), url=rng.choice(words) + rng.choice(words) + rng.choice(words), path=rng.choice(words) + rng.choice(words),
                    method=rng.choice(["GET", "POST", "PUT"]), user_agent="ua", remote_port=rng.randint(1, 9), asn=rng.choice([64, 65]), country=rng.choice(["FR", "US"]))
            for _ in range(5000)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, "residual rules on the device")
    assert len(set(want["rule_idx"].tolist())) >= 6
    eng.close()


@pytest.mark.parametrize("how", list(JIT))
def test_execution_errors_are_counted_per_rule(how):
    """The reference logs every rule whose execution errs (pingoo/rules.rs:41-45: warn!, no match). Run-time errors only arise on the
    per-request interpreter (checked arithmetic, computed indexes): the device counts them per caller rule, across batches, and the
    counts equal the oracle's."""
    rules = [("div", "client.remote_port / (client.remote_port - 80) == 1", [B]),            # division by zero for port 80
             ("ovf", "9223372036854775807 + client.remote_port > 0", [B]),                   # overflow unless the port is 0
             ("idx", '[http_request.host, http_request.path][client.remote_port - 80] == "h"', [B]),  # index out of range unless the port is 80 or 81
             ("fine", "http_request.path.length() + 1 > http_request.url.length()", [CAP]),
             ("col", 'http_request.path == "/x"', [B])]
    eng = RuleEngine(rules, flags=JIT[how])
    check_mode(eng, how)
    orc = pyoracle.Oracle(rules, flags=0)
    rng = random.Random(8)
    reqs = [Request(host="h", path=rng.choice(["/x", "/y"]), url="/y?z=1", user_agent="ua", remote_port=rng.choice([0, 80, 81, 82, 443])) for _ in range(4000)]
    batch = RequestBatch.from_requests(reqs)
    want = orc.evaluate(batch)
    for rounds in (1, 2):
        H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, "errors are no-matches")
        errs = eng.rule_errors(len(rules))
        # the oracle's view: a rule errs for a request when execute_rule says 3 (only for requests that REACH it is what a sequential
        # host would log; the device evaluates every residual rule for every request, so the count is over all requests)
        expect = [sum(orc.execute_rule(k, batch, i) == 3 for i in range(batch.n)) * rounds for k in range(len(rules))]
        assert errs == expect and errs[0] > 0 and errs[1] > 0 and errs[2] > 0 and errs[3] == 0 and errs[4] == 0, (errs, expect)
    eng.close()


@pytest.mark.parametrize("how", list(JIT))
def test_more_than_32_residual_rules(how):
    """The specialized program writes one result word per request and 32 rules; the verdict kernel reads the first word one group ahead
    and the others in the group: 70 residual rules = three words, some rules erring at run time (counted per rule), batch sizes around
    the 64-request group."""
    rules = []
    for k in range(70):
        if k % 9 == 4:
            e = f"{k} / (client.remote_port % 3 - 1) >= 0 && client.remote_port % 70 == {k}"  # divides by zero for a third of the ports: an execution error
        elif k % 2:
            e = f"client.remote_port % 70 == {k} && http_request.path.length() + {k} > http_request.url.length()"
        else:
            e = f'(http_request.path + "{k}").ends_with("{k % 10}") && client.remote_port % 70 == {k}'
        rules.append((f"r{k}", e, [B] if k % 3 else [CAP]))
    eng = RuleEngine(rules, flags=JIT[how] | _abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
    check_mode(eng, how)
    orc = pyoracle.Oracle(rules, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
    rng = random.Random(70)
    total_err = [0] * len(rules)
    for n in (1, 63, 64, 65, 1000):
        reqs = [Request(host="h", path="/" + "p" * rng.randint(0, 12), url="/" + "u" * rng.randint(0, 40), user_agent="ua", remote_port=rng.randint(0, 20000)) for _ in range(n)]
        batch = RequestBatch.from_requests(reqs)
        want = orc.evaluate(batch)
        H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, f"70 residual rules, n={n}")
        for k in range(len(rules)):
            total_err[k] += sum(orc.execute_rule(k, batch, i) == 3 for i in range(n))
    assert len(set(want["rule_idx"].tolist())) > 20  # rules of every word decide requests
    assert eng.rule_errors(len(rules)) == total_err and sum(total_err) > 0
    eng.close()


@pytest.mark.parametrize("how", list(JIT))
def test_computed_keys_into_http_request_and_headers_on_the_device(how):
    """Round 5: http_request[k] / http_request.headers[k] with a computed key (the headers map's names: the rule set's literal keys,
    closed before any rule is compiled), the rule set of tests/test_residual.py's known answers plus column rules around them."""
    rules = [("field", 'http_request["pa" + (http_request.method.length() % 1 == 0 ? "th" : "x")].starts_with("/adm")', [B]),
             ("which", 'http_request[http_request.method == "GET" ? "host" : "path"] == "a.example"', [CAP]),
             ("hname", 'http_request.headers["x-" + http_request.headers.cookie] == "1"', [B]),
             ("member", '(http_request.host + "") in http_request.headers', [CAP]),
             ("count", 'http_request.headers.length() == 2 && http_request.path == "/count"', [B]),
             ("via", 'http_request["headers"][http_request.path] == "1"', [CAP]),
             ("absent", 'http_request["nope" + http_request.method] == "x"', [B]),
             ("plain", '"x-a" in http_request.headers && http_request.path.contains("plain")', [B])]
    eng = RuleEngine(rules, flags=JIT[how])
    check_mode(eng, how)
    assert not eng.partial and eng.header_names == ["cookie", "x-a"]
    rng = random.Random(11)
    words = ["/admin", "a.example", "x-a", "cookie", "/count", "/plain", "a", "1", "", "/"]
    reqs = [Request(host=rng.choice(words), url="/" + rng.choice(words), path=rng.choice(words), method=rng.choice(["GET", "POST", "PUT"]), user_agent="ua",
                    headers={h: rng.choice(["1", "a", "", "2"]) for h in ("x-a", "cookie") if rng.random() < 0.8})
            for _ in range(3000)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, "computed keys on the device")
    assert len(set(want["rule_idx"].tolist())) >= 6, set(want["rule_idx"].tolist())
    eng.close()


@pytest.mark.parametrize("how", list(JIT))
def test_wide_header_maps_computed_patterns_and_reordered_chains_on_the_device(how):
    """Round 6: (a) a rule set that mentions 64 header names and needs the headers map as a value (a constant of the program: one stack
    slot — such rules were refused by name), (b) `matches` with a pattern computed from finitely many strings (every candidate compiled
    at creation, the value picks its table), (c) && / || chains of pure operands evaluated cheapest first, one with an operand that can
    fail kept in source order — error counts included."""
    names = [f"x-h{k}" for k in range(64)]
    lists = dict(TR.LISTS, pats=(_abi.LIST_STRING, ["^/adm", "\\.php$", "(", "^[a-z]+$"]))
    rules = [(f"h{k}", f'http_request.headers["{n}"] == "{k}" && http_request.path == "/h{k}"', [B]) for k, n in enumerate(names)]
    rules += [("byname", 'http_request.headers[http_request.method] == "v"', [CAP]),
              ("member", '(http_request.host + "") in http_request.headers && http_request.path == "/m"', [B]),
              ("via", 'http_request["headers"][http_request.host] == "7"', [CAP]),
              ("pat_cond", 'http_request.path.matches(http_request.method == "GET" ? "^/adm" : "^/api")', [B]),
              ("pat_list", 'http_request.path.matches(lists["pats"][client.remote_port % 4])', [CAP]),
              ("pat_cat", 'http_request.url.matches("^/" + (client.remote_port > 100 ? "a" : "b") + "[a-z]*$")', [B]),
              ("chain", '(http_request.host + ":" + http_request.method).matches("^[a-z]+:(GET|POST)$") && http_request.path + "x" == "/qx"', [B]),
              ("chain_err", 'http_request.url.contains(http_request.host) && client.remote_port / (client.remote_port - 80) == 1', [CAP]),
              ("chain_or", 'http_request.url.contains(http_request.host + "!") || http_request.path.length() * 2 + 1 == http_request.url.length()', [B])]
    eng = RuleEngine(rules, lists, flags=JIT[how])
    check_mode(eng, how)
    assert not eng.partial and eng.header_names == names
    rng = random.Random(66)
    reqs = []
    for _ in range(4000):
        hd = {n: rng.choice([str(k), "v", "7", ""]) for k, n in enumerate(names) if rng.random() < 0.3}
        reqs.append(Request(host=rng.choice(["x-h3", "x-h40", "ab", "x-h64", "a.example"]), path=rng.choice(["/h5", "/h40", "/m", "/admin", "/api/x.php", "/q", "/b", "/abc"]),
                            url=rng.choice(["/ab", "/b", "/abc?ab", "/q/ab!", "/qq1"]), method=rng.choice(["GET", "POST", "x-h12", "x-h7"]), user_agent="ua",
                            remote_port=rng.choice([0, 1, 2, 3, 80, 101, 443]), headers=hd))
    batch = RequestBatch.from_requests(reqs)
    orc = pyoracle.Oracle(rules, lists)
    want = orc.evaluate(batch)
    H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, "64 header names, computed patterns, reordered chains")
    assert len(set(want["rule_idx"].tolist())) >= 12, set(want["rule_idx"].tolist())
    n_res = [k for k, r in enumerate(rules) if any(f"rule #{k} " in w and "residual" in w for w in eng.program.warnings())]
    errs = eng.rule_errors(len(rules))
    for k in n_res:  # execution errors per residual rule = the oracle's (every residual rule is evaluated for every request)
        assert errs[k] == sum(orc.execute_rule(k, batch, i) == 3 for i in range(batch.n)), (rules[k][0], errs[k])
    assert len(n_res) >= 8 and errs[len(names) + 4] > 0 and errs[len(names) + 7] > 0  # (the invalid list pattern for a quarter of the ports; the division by zero at port 80)
    eng.close()
