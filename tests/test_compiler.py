"""Host compiler (no GPU): front-end accept/reject parity with the oracle, and the compiled tables —
interpreted by the test-only walker in tests/table_walker.py — against the oracle's verdicts."""
import random

import numpy as np
import pytest

import helpers as H
from oracle import pyoracle
from pingoo_amd import Request, RequestBatch, _abi
from pingoo_amd.engine import CompiledProgram, ExpressionIsNotValid, PwafError, UnsupportedExpression, compile_expression, validate_expression
from table_walker import Tables

B, CAP = _abi.RULE_ACTION_BLOCK, _abi.RULE_ACTION_CAPTCHA


def walk(program: CompiledProgram, batch: RequestBatch):
    t = Tables(program)
    out = np.zeros(batch.n, dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    for i in range(batch.n):
        out[i] = t.evaluate(batch, i)
    return out


def test_golden_vectors_through_compiled_tables(kat):
    for c in kat["cases"]:
        rules, lists, batch, expect = H.kat_case_inputs(c)
        got = walk(CompiledProgram(rules, lists), batch)
        assert [(int(v["action"]), int(v["rule_idx"])) for v in got] == [tuple(e) for e in expect.tolist()], c["name"]


def test_syntax_accept_reject_parity_with_oracle():
    """Two independently written parsers (recursive descent vs Pratt) must agree on every input."""
    rng = random.Random(2024)
    toks = ["a", "http_request", ".", "path", "(", ")", "[", "]", "{", "}", ",", ":", "?", "!", "-", "+", "*", "/", "%", "==", "!=", "<", "<=", ">", ">=", "&&", "||", "in", "true",
            "null", "1", "2.5", "0x1f", '"s"', "'t'", 'r"\\d"', " ", "contains", "9223372036854775808", "1u", '"\\q"', "//c\n", "\n", ".5", "1e3", "e", "_x1", "@", '"', "-9223372036854775808"]
    n_ok = n_bad = 0
    for _ in range(6000):
        e = "".join(rng.choice(toks) + rng.choice(["", "", " "]) for _ in range(rng.randint(1, 9)))
        try:
            pyoracle.compile_expression(e)
            o_ok = True
        except pyoracle.OracleError:
            o_ok = False
        try:
            compile_expression(e)
            p_ok = True
        except ExpressionIsNotValid:
            p_ok = False
        assert o_ok == p_ok, repr(e)
        try:
            pyoracle.validate_expression(e)
            ov = True
        except pyoracle.OracleError:
            ov = False
        try:
            validate_expression(e)
            pv = True
        except ExpressionIsNotValid:
            pv = False
        assert ov == pv, repr(e)
        n_ok += o_ok
        n_bad += not o_ok
    assert n_ok > 300 and n_bad > 300, (n_ok, n_bad)
    for e in [H.rexpr(rng, {}) for _ in range(300)]:
        pyoracle.compile_expression(e)
        compile_expression(e)


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_compiled_tables_match_oracle(seed):
    rng = random.Random(1000 + seed)
    lists = H.fuzz_lists(rng)
    geo = H.fuzz_geoip(rng) if rng.random() < 0.7 else None
    with_geo = rng.random() < 0.3
    rules = []
    for k in range(rng.randint(1, 12)):
        e = H.rexpr(rng, lists) if rng.random() < 0.95 else None
        acts = H.fuzz_actions(rng)
        rules.append((f"r{k}", e, acts))
    flags = rng.choice([0, 0, _abi.OPT_NO_UA_GATE, _abi.OPT_NO_CAPTCHA_BYPASS, _abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS])
    prog = CompiledProgram(rules, lists, geo, flags=flags | _abi.OPT_LENIENT, max_table_bytes=rng.choice([0, 0, 2048, 4096]), max_dfa_states=rng.choice([0, 0, 40]))
    # (lenient only so that a rule NOBODY can take would show up as a count instead of an exception: the grammar produces none —
    # what the column compiler cannot take runs in the residual interpreter, evaluated here by its host build)
    seen, _ = H.as_the_engine_sees(rules, prog)
    batch = RequestBatch.from_requests(H.fuzz_requests(rng, 48, with_geo))
    want = pyoracle.Oracle(seen, lists, geo, flags=flags).evaluate(batch)
    H.assert_verdicts_equal(walk(prog, batch), want, batch, f"seed {seed}")


def test_synthetic_workload_tables_match_oracle():
    from synth import pysynth

    for cid, n in ((0, 1500), (2, 400)):
        w = pysynth.Workload(cid)
        prog = CompiledProgram(w.rules, w.lists, w.geoip)
        assert not prog.warnings(), prog.warnings()
        # take the requests that do NOT end up Allow plus a slice of ordinary ones, so rule paths are exercised
        big = w.batch(0, 40000)
        ov = pyoracle.Oracle(w.rules, w.lists, w.geoip).evaluate(big, threads=4)
        idx = np.concatenate([np.nonzero(ov["action"] != 0)[0][: n // 2], np.arange(n // 2)])
        reqs = RequestBatch(
            [np.concatenate([big.data[f][big.offsets[f][i]:big.offsets[f][i + 1]] for i in idx] + [np.zeros(16, np.uint8)]) for f in range(5)],
            [np.concatenate([[0], np.cumsum([int(big.offsets[f][i + 1]) - int(big.offsets[f][i]) for i in idx])]).astype(np.uint32) for f in range(5)],
            big.ip[idx], big.ip_is_v6[idx], big.port[idx], big.flags[idx])
        want = pyoracle.Oracle(w.rules, w.lists, w.geoip).evaluate(reqs)
        assert (want["action"] != 0).sum() >= min(n // 2, 50)
        H.assert_verdicts_equal(walk(prog, reqs), want, reqs, f"workload {cid}")


def test_static_errors_become_warnings_not_failures():
    p = CompiledProgram([("a", "http_request.nope == 1", [B]), ("b", "http_request.path", [B]), ("c", 'http_request.path.matches("(")', [B]),
                         ("d", 'lists["missing"].contains(client.ip)', [B]), ("e", "undefined(1)", [B]), ("ok", 'http_request.path == "/x"', [B])])
    w = p.warnings()
    assert len(w) == 5 and all("never match" in x for x in w), w
    assert p.stats()["n_rules"] == 3  # UA gate, captcha endpoint, and the one live rule


def test_unsupported_constructs_are_rejected_with_the_rule_index():
    # what neither the column compiler nor the residual interpreter takes: a regex compiled per request, Unicode properties beyond
    # categories / scripts / the Perl classes' ingredients, CRLF mode, the word-EDGE assertions (scripts and (?x) are taken since round 5)
    cases = ['http_request.path.matches(http_request.host)', 'http_request.url.matches("\\\\p{Age=6.0}")', 'http_request.path.matches("(?Rm)a$")',
             'http_request.path.matches("\\\\b{start}a")']  # (http_request[computed] and computed header names: taken since round 5, tests/test_residual.py)
    for e in cases:
        pyoracle.compile_expression(e)  # valid language, just outside what the device evaluates
        with pytest.raises(UnsupportedExpression) as ei:  # the default since ABI 2: creation fails, naming the rule
            CompiledProgram([("fine", 'http_request.path == "/"', [B]), ("bad", e, [B])])
        assert ei.value.rule_index == 1 and "bad" in ei.value.message, e
        # PWAF_OPT_LENIENT: that rule alone is reported and never matches; the set compiles (with a warning status)
        prog = CompiledProgram([("fine", 'http_request.path == "/"', [B]), ("bad", e, [B])], flags=_abi.OPT_LENIENT)
        assert prog.partial
        assert prog.rule_status(0) == (0, "") and prog.rule_status(1)[0] == _abi.E_UNSUPPORTED and "bad" in prog.rule_status(1)[1]
        assert any("NOT evaluated" in w for w in prog.warnings())
    # what only the column compiler cannot take runs in the residual interpreter: no error, a warning says so
    residual = ['client.country == http_request.host', "client.remote_port + 1 == 81", 'http_request.path < "m"',
                '(http_request.method == "GET" ? http_request.path : http_request.url) == "/"', 'http_request.path + "x" == "/x"', "[http_request.method].contains(\"GET\")"]
    for e in residual:
        prog = CompiledProgram([("fine", 'http_request.path == "/"', [B]), ("slow", e, [B])])
        assert not prog.partial and prog.unsupported_rules(2) == []
        assert any("residual interpreter" in w for w in prog.warnings()), (e, prog.warnings())
        with pytest.raises(UnsupportedExpression):
            CompiledProgram([("slow", e, [B])], flags=_abi.OPT_NO_RESIDUAL)
    with pytest.raises(ExpressionIsNotValid) as ei:
        CompiledProgram([("x", "a ==", [B])])
    assert ei.value.rule_index == 0
    with pytest.raises(PwafError) as ei:
        CompiledProgram([("x", None, [7])])
    assert ei.value.code == _abi.E_INVALID_ARG
    with pytest.raises(PwafError) as ei:
        CompiledProgram([("x", None, [B])], {"l": (_abi.LIST_IP, ["1.2.3"])})
    assert ei.value.code == _abi.E_LIST and "line 1" in ei.value.message


def test_dfa_grouping_respects_the_lds_budget_and_keeps_results():
    rng = random.Random(7)
    words = ["".join(rng.choice("abcdefgh") for _ in range(rng.randint(3, 7))) for _ in range(120)]
    rules = [(f"r{k}", f'http_request.path.contains("{w}")', [B]) for k, w in enumerate(words)]
    big = CompiledProgram(rules)
    small = CompiledProgram(rules, max_table_bytes=4096)
    assert big.stats()["n_dfa_groups"] == 1  # one table per field: the literals and the captcha-endpoint prefix share the path table
    assert small.stats()["n_dfa_groups"] > big.stats()["n_dfa_groups"]
    t = Tables(small)
    for g in t.groups:
        assert g["n_states"] * g["n_classes"] * 2 <= 4096
    reqs = [Request(path="/" + "".join(rng.choice(words + ["zz", "/"]) for _ in range(rng.randint(0, 3)))) for _ in range(200)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    H.assert_verdicts_equal(walk(big, batch), want, batch, "one group")
    H.assert_verdicts_equal(walk(small, batch), want, batch, "split groups")


def test_atoms_are_shared_across_rules():
    rules = [(f"r{k}", 'http_request.path.contains("/admin") && client.remote_port == %d' % k, [B]) for k in range(50)]
    s = CompiledProgram(rules).stats()
    assert s["n_scan_atoms"] == 2  # "/admin" once + the captcha-endpoint prefix
    assert s["n_numeric_atoms"] == 50 + 2  # 50 ports + the two UA-length gate atoms


COUNTED_GAPS = ['select.{0,40}from', 'a.{0,40}b', '<script[^>]{0,64}>', 'union\\\\s{1,8}select', '(a|b).{2,12}c$', 'x.{0,5}x.{0,5}x', 'ab.{1,9}ab', '^/p.{3,30}\\\\.php',
                '(?i)on\\\\w{0,12}\\\\s{0,4}=', 'select.{0,24}from.{0,24}where', '(?s)a.{0,33}b', 'q{2,13}-', '(select|union).{0,30}(from|where)']


def test_counted_gap_patterns_compile_to_small_tables_and_match_the_oracle():
    """`a.{0,n}b` (the common WAF signature shape; the reference's regex crate takes it, rules/rules.rs compile path): the subset
    construction keeps one thread per counted-class chain (dfa.cpp prune_core), so the table is polynomial in n; verdicts against the oracle's
    backtracking matcher on inputs built from the patterns' own pieces (gap lengths on both sides of every bound, newlines inside gaps)."""
    rules = [(f"r{k}", f'http_request.url.matches("{p}")', [B]) for k, p in enumerate(COUNTED_GAPS)]
    prog = CompiledProgram(rules)
    assert prog.unsupported_rules(len(rules)) == []
    rng = random.Random(1)
    pieces = ["select", "from", "where", "union", "a", "b", "c", "x", "ab", "<script", ">", " ", "\n", "=", "on", "load", "/p", ".php", "-" * 7, "q" * 13, "q", "A", "SeLeCt", "z" * 11, "z" * 29]
    reqs = [Request(url="".join(rng.choice(pieces) for _ in range(rng.randrange(0, 12))), path="/", host="h") for _ in range(6000)]
    for n in (39, 40, 41, 42):   # exact bounds of the first two patterns
        reqs += [Request(url="select" + "y" * n + "from"), Request(url="a" + "y" * n + "b"), Request(url="a" + "y" * (n // 2) + "\n" + "y" * (n - n // 2 - 1) + "b")]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    H.assert_verdicts_equal(walk(prog, batch), want, batch, "counted gaps")
    assert len(set(want["rule_idx"][want["action"] > 0].tolist())) >= 10


def test_a_pattern_beyond_the_table_budget_fails_alone():
    rules = [("big", 'http_request.url.matches("select.{0,60}from.{0,60}where")', [B]), ("min", 'http_request.url.matches("a.{20,40}b")', [B]), ("ok", 'http_request.url.contains("x")', [CAP])]
    prog = CompiledProgram(rules, flags=_abi.OPT_LENIENT)
    assert prog.unsupported_rules(len(rules)) == [0, 1] and "budget" in prog.rule_status(0)[1]
    with pytest.raises(UnsupportedExpression):
        CompiledProgram(rules)


@pytest.mark.parametrize("seed", range(12))
def test_random_counted_repetitions_match_the_oracle(seed):
    """Random patterns around counted repetitions of byte classes (the chains dfa.cpp prune_core keeps one thread of): several chains per
    pattern, chains inside groups and alternations, anchors, min > 0 — against the oracle's backtracking matcher on short random strings
    over the patterns' own alphabet."""
    rng = random.Random(4200 + seed)
    alpha = "abc-"
    def cls():
        return rng.choice([".", "[ab]", "[^a]", "\\\\w", "[a-c]", "[^-]"])
    def piece(depth=0):
        r = rng.random()
        if r < 0.35:
            return rng.choice(alpha[:3]) * rng.randint(1, 2)
        if r < 0.75:
            lo = rng.choice([0, 0, 0, 1, 2])
            return f"{cls()}{{{lo},{lo + rng.randint(2, 9)}}}"
        if r < 0.85 and depth < 2:
            return "(" + "|".join("".join(piece(depth + 1) for _ in range(rng.randint(1, 2))) for _ in range(2)) + ")"
        return rng.choice(["-", "a?", "b+", "c*"])
    pats = []
    for _ in range(10):
        p = "".join(piece() for _ in range(rng.randint(2, 4)))
        pats.append(rng.choice(["", "^"]) + p + rng.choice(["", "", "$"]))
    rules = [(f"r{k}", f'http_request.url.matches("{p}")', [B]) for k, p in enumerate(pats)]
    prog = CompiledProgram(rules, flags=_abi.OPT_LENIENT)
    seen, bad = H.as_the_engine_sees(rules, prog, allow=3)
    reqs = [Request(url="".join(rng.choice(alpha) for _ in range(rng.randint(0, 14)))) for _ in range(400)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(seen).evaluate(batch)
    H.assert_verdicts_equal(walk(prog, batch), want, batch, f"seed {seed}: {pats}")
    assert len(bad) <= 3
    for k, rule in enumerate(rules):  # and every pattern on its own (first-match-wins hides the later ones above)
        if k in bad:
            continue
        one = CompiledProgram([rule])
        H.assert_verdicts_equal(walk(one, batch), pyoracle.Oracle([rule]).evaluate(batch), batch, f"seed {seed}: {pats[k]}")


def test_unicode_classes_follow_the_regex_crate():
    """regex 1.12.2 (Cargo.lock:1694-1700) is Unicode-aware by default and url / path reach it as Rust str that may hold UTF-8 (http
    1.3.1, Cargo.lock:824-826; VERDICT r4 missing #2): `.` and negated classes take one scalar value, \\d \\s \\w \\b and \\p{..} read the
    Unicode tables, (?i) is simple case folding, (?-u) brings ASCII back, (?x) is syntax. The device compiler's UTF-8 automata (CPU
    walk of the compiled tables) against the oracle, pattern by pattern and all in one DFA; (?i) folds a category before \\P / {^..}
    negates it (regex-syntax's order)."""
    E, EU, NB, EM = "\u00e9", "\u20ac", "\u00a0", "\U0001F600"
    cases = [(r"^\p{L}+\p{N}$", ["abc7", "abc", "7", "aB9", "caf" + E + "\u0663", E + EU]), (r"^\P{L}+$", ["123-_", "12a", "", EU + "1", E]), (r"(?i)^\p{Lu}+$", ["abC", "ab1", E + "\u00c9"]),
             (r"^\p{Lu}\p{Ll}+$", ["Abc", "abc", "ABc", "\u00c9" + E]), (r"[\p{N}x]{3}", ["a1x2", "a1b2", "\u0663x\u00b2"]), (r"^\pL\pN$", ["a1", "1a"]), (r"^\p{P}+$", ["!?.-(", "!$", "\u00a1\u2014"]),
             (r"^\p{^N}$", ["7", "x", E]), (r"(?i)^[\P{Lu}]$", ["a", "-"]), (r"^[\P{Lu}]$", ["a", "A", "\u00c9", E]), (r"\p{S}\p{Zs}", ["a+ b", "a+b", EU + NB]),
             (r"id=\p{Nd}{3,}\P{Nd}", ["x?id=1234&", "x?id=12&", "id=\u0663\u0664\u0665" + E]),
             (r"(?i)union\s+select", ["q=union" + NB + "select", "q=UNION\u2003\u3000SELECT", "q=union+select", "union\u200bselect"]),
             (r"(?i)select", ["\u017felect", "SELECT", "\u017eelect"]), (r"(?i)nikto", ["ni\u212ato", "NIKTO", "ni\u212bto"]), (r"select", ["\u017felect", "xselectx"]),
             (r"^\w+$", ["caf" + E, "\u4f60\u597d_1", "a" + EU + "b", "a\u0301", "\u200d"]), (r"^a.b$", ["a" + E + "b", "a" + EM + "b", "a\nb", "ab", "a" + E + E + "b"]),
             (r"^a[^x]b$", ["a" + EU + "b", "axb", "a" + EM + "b"]), (r"^.{3}$", [E + EU + EM, E + EU, "abc", "ab"]), (r"^\d+$", ["\u0663\u0967", "\u00b2", "12"]),
             (r"^\D$", ["\u00b2", "\u0663", "x"]), (r"^\S$", [E, "\u2028", " "]), (r"^\W$", [EU, E, "-"]), (r"\bselect", [E + "select", EU + "select", "select", " select", "xselect"]),
             (r"select\b", ["select" + E, "select" + EU, "select", "selectx", "select\u2003x"]), (r"select\B", ["select" + E, "select" + EU, "select", "selectx"]),
             (r"(?-u:\b)select", [E + "select", "xselect", "-select"]), (r"\Bsel\b.\bect", ["xsel-ect", "xsel" + E + "ect", "sel-ect", E + "sel" + EU + "ect"]),
             (r"x\b.*\by", ["x-y", "x" + E + "y", "x" + E + " y", "x " + E + "y", "x " + E + " y"]), (r"\b" + E + r"t\b", ["caf" + E + "t", "caf " + E + "t", E + "t" + E, EU + E + "t" + EU]),
             (r"(?-u:\w)$", [E, "a"]), (r"(?i-u)select", ["\u017felect", "SELECT"]), (r"(?-u:\s)x", [NB + "x", " x"]),
             ("^[\u03b1-\u03c9]+$", ["\u03b1\u03c9", "\u0391"]), ("(?i)^[\u03b1-\u03c9]+$", ["\u0391\u03a9", "a"]), (r"(?i)^[[:lower:]]$", ["\u212a", "\u017f", "k", E]), (r"^[[:^alpha:]]$", [E, "a", "1"]),
             (r"^\x{e9}\u00e9\u{e9}\U000000e9\xe9$", [E * 5, E * 4]), (r"^\p{Greek}+$", ["\u03b1\u03b2", "\u03b1a"]), (r"^\p{sc=Cyrillic}\p{Script=Latin}$", ["\u0436z", "z\u0436"]),
             (r"^\p{gc=Nd}\p{Alphabetic}\p{White_Space}$", ["\u0663" + E + "\u3000", "1a "]), (r"(?x) union \s+ select  # comment", ["union select", "unionselect"]),
             (r"(?x)a\ b [ c d ]{ 1, 2 }$", ["a bdc", "a b d"]), (r"(?s)^.$", ["\n", E]), (r"^[^\n" + E + "]+$", ["a" + EU, "a" + E]), (r"(?i)\u017f\u212a", ["sk", "SK", "\u017fK"]),
             (r"<script[^>]*>", ["<script " + E + EM + ">", "<script " + E], ),
             # without the u flag only the fixed two-digit \xHH is a BYTE (regex-syntax: Literal::byte() is Some for HexFixed(X) alone); the brace, \u and \U
             # forms denote the scalar value = its UTF-8 encoding, like a raw non-ASCII character
             (r"(?-u)caf\x{e9}$", ["caf" + E, "cafe"]), (r"(?-u:\u00e9\U000000e9)" + E, [E * 3, E * 2]), (r"(?i-u)\x{e9}x", [E + "X", "\u00c9x"])]
    rules = [(f"r{k}", f"http_request.path.matches({H.q(pat)})", [H.B]) for k, (pat, _) in enumerate(cases)]
    prog = CompiledProgram(rules, {}, flags=_abi.OPT_NO_UA_GATE)
    assert prog.unsupported_rules(len(rules)) == []
    t = Tables(prog)
    n_match = 0
    for k, (pat, hays) in enumerate(cases):
        orc = pyoracle.Oracle([rules[k]], {}, flags=_abi.OPT_NO_UA_GATE)
        one = Tables(CompiledProgram([rules[k]], {}, flags=_abi.OPT_NO_UA_GATE))
        for h in hays:
            batch = RequestBatch.from_requests([Request(path=h, url="/", host="h")])
            want = orc.evaluate(batch)[0]
            assert pyoracle.regex_is_match(pat, h.encode()) == (int(want["action"]) == 1), (pat, h)
            assert one.evaluate(batch, 0) == (int(want["action"]), int(want["rule_idx"])), (pat, h)
            n_match += int(want["action"]) == 1
    assert n_match > 60
    # every pattern in ONE table against the oracle's first match
    hays = sorted({h for _, hs in cases for h in hs})
    batch = RequestBatch.from_requests([Request(path=h, url="/", host="h") for h in hays])
    want = pyoracle.Oracle(rules, {}, flags=_abi.OPT_NO_UA_GATE).evaluate(batch)
    for i in range(batch.n):
        assert t.evaluate(batch, i) == (int(want[i]["action"]), int(want[i]["rule_idx"])), hays[i]
    for bad in (r"\p{Age=6.0}", r"\p{scx=Latin}", r"\p{Emoji}"):
        with pytest.raises(UnsupportedExpression):
            CompiledProgram([("g", f"http_request.path.matches({H.q(bad)})", [H.B])], {})
    # an unterminated property, or what the crate's UTF-8 mode refuses under (?-u), is an INVALID pattern: an execution error in the
    # reference, i.e. a rule that never matches (D14)
    for inv in (chr(92) + "p{", "(?-u:.)", r"(?-u:\W)", r"(?-u:[^a])", r"(?-u:\xFF)", r"(?-u:\pL)", r"\x{D800}"):
        assert any("never match" in w for w in CompiledProgram([("g", f"http_request.path.matches({H.q(inv)})", [H.B])], {}).warnings()), inv
        with pytest.raises(pyoracle.OracleError):
            pyoracle.regex_is_match(inv, b"a")


def test_absolute_form_urls_http2_stream():
    """`url` is Display(Uri) (pingoo/serde_utils.rs:16-18): the origin form on HTTP/1, the absolute form (https://host/path?query) on
    HTTP/2. The 1k-rule set on the same stream with absolute urls: rules anchored on the url's first bytes stop matching, rules on
    `path` do not care — and the compiled tables agree with the oracle on both (VERDICT r3 missing #5)."""
    from synth import pysynth

    w = pysynth.Workload(3)
    prog = CompiledProgram(w.rules, w.lists, w.geoip)
    t = Tables(prog)
    orc = pyoracle.Oracle(w.rules, w.lists, w.geoip)
    differ = 0
    a, b = w.batch(40_000, 250), w.batch(40_000, 250, absolute_url=True)
    wa, wb = orc.evaluate(a, threads=8), orc.evaluate(b, threads=8)
    for i in range(a.n):
        assert b.field_bytes(1, i).startswith(b"https://" + b.field_bytes(0, i)) and a.field_bytes(2, i) == b.field_bytes(2, i)
        assert t.evaluate(a, i) == (int(wa[i]["action"]), int(wa[i]["rule_idx"])), i
        assert t.evaluate(b, i) == (int(wb[i]["action"]), int(wb[i]["rule_idx"])), i
        differ += int(wa[i]["rule_idx"]) != int(wb[i]["rule_idx"])
    assert np.count_nonzero(wb["action"]) > 0


def test_utf8_stream_of_the_1k_rule_set():
    """The synthetic stream with UTF-8 in url / path (synth mode 4: segments in other scripts, `union<U+00A0>select`, U+017F for s,
    near-boundary e-acute / euro signs) through the 1k-rule set: the compiled tables (UTF-8 automata, the \\b rewrite, the bigram filter's
    factors) agree with the oracle request by request, and the Unicode-only matches are there (VERDICT r4 missing #2)."""
    from synth import pysynth

    w = pysynth.Workload(3)
    prog = CompiledProgram(w.rules, w.lists, w.geoip)
    prog.tune(w.batch(9_000_000, 4096))  # (filters as the bench runs them; benign ASCII sample)
    t = Tables(prog)
    orc = pyoracle.Oracle(w.rules, w.lists, w.geoip)
    a, b = w.batch(60_000, 400), w.batch(60_000, 400, utf8=True)
    wa, wb = orc.evaluate(a, threads=8), orc.evaluate(b, threads=8)
    non_ascii = 0
    for i in range(b.n):
        b.field_bytes(1, i).decode("utf-8"), b.field_bytes(2, i).decode("utf-8")  # well-formed, as a Rust str is
        non_ascii += not b.field_bytes(1, i).isascii()
        assert t.evaluate(b, i) == (int(wb[i]["action"]), int(wb[i]["rule_idx"])), (i, b.field_bytes(1, i))
    assert non_ascii > 150 and np.count_nonzero(wb["action"]) > np.count_nonzero(wa["action"]) + 10


def test_field_against_field_predicates_beyond_the_device_table_become_residual_rules():
    """The device evaluates field-against-field atoms in one pseudo pass of 32 predicates over 8 distinct fields. A rule set that needs more
    used to fail creation as a whole (PWAF_E_UNSUPPORTED without a rule index, lenient or not: found by tools/headerfuzz.py, round 5); the
    rules beyond the table are now lowered to residual programs — same verdicts, by the reference's rule: every valid expression is
    evaluated (pingoo/rules.rs:37-51)."""
    import numpy as np
    import table_walker
    from pingoo_amd import Request, RequestBatch
    from pingoo_amd.engine import CompiledProgram

    names = [f"x-h{k}" for k in range(12)]
    rules = [(f"r{k}", f'http_request.headers["{names[k % 12]}"] {["==", "!="][k % 2]} http_request.{["host", "path", "method", "url"][k % 4]}' +
              (f' && http_request.headers["{names[(k + 5) % 12]}"].contains(http_request.host)' if k % 3 == 0 else ""), [H.B if k % 2 else H.CAP]) for k in range(48)]
    prog = CompiledProgram(rules)  # strict: must not raise
    assert sorted(prog.header_names) == sorted(names) and prog.header_names == pyoracle.Oracle(rules).header_names  # (order of first use)
    n_res = sum("residual" in w for w in prog.warnings())
    assert 8 <= n_res < 48, n_res  # the table's share stays on the column path, the rest is residual
    rng = random.Random(3)
    reqs = [Request(host=rng.choice(["a", "b", ""]), path="/" + rng.choice(["a", "b"]), url="/" + rng.choice(["a", "b"]), method=rng.choice(["a", "GET"]), user_agent="ua",
                    headers={nm: rng.choice(["a", "b", "/a", "/b", "GET", ""]) for nm in names if rng.random() < 0.7}) for _ in range(200)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    t = table_walker.Tables(prog)
    got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    H.assert_verdicts_equal(got, want, batch, "field-against-field overflow")
    assert len(set(want["rule_idx"].tolist())) >= 4


@pytest.mark.parametrize("what", ["asn_comparisons", "header_lengths", "country_tables", "port_sets"])
def test_rule_sets_beyond_a_device_table_width_fall_to_residual_programs(what):
    """Widths of the attribute kernel's rows (128 asn comparisons, 8 header-length variables, 256 country tables, 128 integer sets per
    client variable) used to fail ENGINE creation as a whole once a rule set crossed one — 200 rules `client.asn == N` are a plausible rule
    set. The rule that would cross a width is now lowered to a residual program (round 5): nothing refused, same verdicts."""
    import table_walker

    rng = random.Random(17)
    if what == "asn_comparisons":
        rules = [(f"r{k}", f"client.asn == {1000 + k}", [H.B]) for k in range(200)]
        n_over = 200 - 128
    elif what == "header_lengths":
        rules = [(f"r{k}", f'http_request.headers["x-h{k}"].length() > {k % 4}', [H.B]) for k in range(12)]
        n_over = 12 - 8
    elif what == "country_tables":
        cc = [chr(65 + a) + chr(65 + b) for a in range(26) for b in range(26)]
        rules = [(f"r{k}", f'["{cc[k]}", "{cc[(7 * k + 3) % 676]}"].contains(client.country) && client.remote_port > 5', [H.B]) for k in range(300)]
        n_over = 300 - 256
    else:
        rules = [(f"r{k}", f"[{k + 2}, {k + 70000 % 60000}, 9].contains(client.remote_port)", [H.B]) for k in range(150)]
        n_over = 150 - 128
    prog = CompiledProgram(rules)  # strict: nothing may be refused
    n_res = sum("residual" in w for w in prog.warnings())
    assert n_over <= n_res < len(rules), (n_res, n_over)
    reqs = [Request(host="h", path="/", url="/", user_agent="ua", remote_port=rng.choice([2, 3, 9, 50, 150, 10000]), asn=rng.choice([1000, 1100, 1150, 1199, 5]),
                    country=rng.choice(["AA", "AD", "KX", "ZZ", "FR"]), headers={f"x-h{k}": "a" * rng.randint(0, 5) for k in range(12) if rng.random() < 0.6}) for _ in range(150)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    t = table_walker.Tables(prog)
    got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    H.assert_verdicts_equal(got, want, batch, what)
    assert len(set(want["rule_idx"].tolist())) >= 3, set(want["rule_idx"].tolist())


ILL_PATTERNS = [r"^a.b$", r"^a.{2}b$", r"(é|x)?ab", r"\bselect", r"select\b", r"^\w+$", r"a[^x]b", r"^(é)+$", r"q=\w+&", r"(?i)café"]
ILL_HAYS = [b"a\x80b", b"ab", b"a\xc3\xa9b", b"a\xc3\xa9\x80b", b"a\xe2\x82b", b"a\xe2\x82\xacb", b"a\xed\xa0\x80b", b"a\xc0\x80b", b"a\xf4\x90\x80\x80b", b"a\xf0\x9f\x98\x80b", b"\x80ab", b"ab\x80",
            b"a\xc3b", b"a\xc3\xa9\xa9b", b"\x80select", b"select\x80", b"\xc3\xa9select", b"select\xbf x", b"\xc3\xa9\xc3\xa9", b"\xc3\xa9\x80\xc3\xa9", b"\xc3\xa9\xc3", b"q=\xc3\xa9\x80&", b"q=\xc3\xa9x&",
            b"caf\xc3\x89", b"caf\xc3\x80\x89", b"a\xbf\xbf\xbfb", b"a\xe2\x82\xac\x80b", b"a\xf0\x9f\x98b", b"\xf0\x9f\x98\x80", b"a\xc3\xa9", b"\xa9b"]


def test_ill_formed_utf8_is_a_unit_no_class_matches_for_every_table_walker():
    """D17 closed: the C ABI takes arbitrary bytes where the reference has Rust str. A byte that belongs to no
    well-formed sequence — a stray continuation byte, a truncated sequence, a surrogate, an overlong form, > U+10FFFF — is a unit that no regex
    class matches, for the oracle's decoder AND for the walkers of a scalar-mode table (they used to SKIP a stray continuation byte:
    `a\\x80b` held "ab"). The compiled tables (CPU walk = csrc/utf8.h restated) against the oracle, pattern by pattern, all patterns in one
    table, and beside byte-substring predicates that share the table."""
    rules = [(f"r{k}", f"http_request.path.matches({H.q(pat)})", [H.B]) for k, pat in enumerate(ILL_PATTERNS)]
    rules += [("lit_ab", 'http_request.path.contains("ab")', [H.B]), ("lit_e", 'http_request.path.ends_with("é")', [H.B]), ("lit_sel", 'http_request.path.starts_with("select")', [H.B])]
    batch = RequestBatch.from_requests([Request(path=h, url=b"/", host="h") for h in ILL_HAYS])
    n_match = 0
    for k, rule in enumerate(rules):
        want = pyoracle.Oracle([rule], {}, flags=_abi.OPT_NO_UA_GATE).evaluate(batch)
        one = Tables(CompiledProgram([rule], {}, flags=_abi.OPT_NO_UA_GATE))
        for i in range(batch.n):
            assert one.evaluate(batch, i) == (int(want[i]["action"]), int(want[i]["rule_idx"])), (rule[1], ILL_HAYS[i])
        n_match += int(np.count_nonzero(want["action"]))
    assert n_match >= 25
    # known answers (hand-derived from decode_units' definition): a stray byte separates, a well-formed sequence is one unit
    orc = pyoracle.Oracle(rules, {}, flags=_abi.OPT_NO_UA_GATE)
    M = lambda pat, h: pyoracle.regex_is_match(pat, h)
    assert not M(r"(é|x)?ab", b"a\x80b") and M(r"^a.b$", b"a\xc3\xa9b") and not M(r"^a.b$", b"a\x80b") and not M(r"^a.{2}b$", b"a\xc3\xa9\x80b") and not M(r"^a.b$", b"a\xe2\x82b")
    assert not M(r"\bselect", b"\x80select") and M(r"\bselect", b" select") and not M(r"^\w+$", b"\xc3\xa9\x80\xc3\xa9") and M(r"^\w+$", b"\xc3\xa9\xc3\xa9")
    # every pattern in ONE table (the literals share it: scalar mode)
    t = Tables(CompiledProgram(rules, {}, flags=_abi.OPT_NO_UA_GATE))
    want = orc.evaluate(batch)
    for i in range(batch.n):
        assert t.evaluate(batch, i) == (int(want[i]["action"]), int(want[i]["rule_idx"])), ILL_HAYS[i]
