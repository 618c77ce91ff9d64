"""Pins the CPU oracle: golden vectors derived from the reference's docs/call sites, and independent
third-party implementations (CPython `re`, `ipaddress`) for the parts whose algorithm lives in
un-vendored dependencies of the reference (regex 1.12.2, ipnetwork 0.21.1, maxminddb 0.24.0)."""
import os
import ipaddress
import random
import re

import numpy as np
import pytest

from helpers import assert_verdicts_equal, kat_case_inputs
from oracle import pyoracle
from pingoo_amd import Request, RequestBatch, _abi, geoip_entries

B, CAP = _abi.RULE_ACTION_BLOCK, _abi.RULE_ACTION_CAPTCHA


def test_golden_vectors(kat):
    for c in kat["cases"]:
        rules, lists, batch, expect = kat_case_inputs(c)
        got = pyoracle.Oracle(rules, lists).evaluate(batch)
        assert [(int(v["action"]), int(v["rule_idx"])) for v in got] == [tuple(e) for e in expect.tolist()], c["name"]


def test_compile_and_validate_accept_reject():
    ok = ['http_request.path == "/x"', "a.b.c", "x in [1, 2,]", "!(!a)", "--1 == 1", 'r"\\d" == "\\\\d"', "a ? b : c ? d : e", "{'k': 1}.k == 1", "-9223372036854775808 < 0",
          "1.5e3 > .5", "0x1F == 31", "a // comment\n && b", '"\\u00e9\\x41\\101" != ""']
    bad = ["", "a ==", "a &&& b", "(a", "a b", "1 +", "a.b(", "[1, 2", '"unterminated', "9223372036854775808", "a ? b", "1u", 'b"x"', "a.", "!-a", "a ? b : ", "@", '"""x"""',
           "Foo{a: 1}", "a[1", "in", "a in", ".a"]
    for e in ok:
        pyoracle.compile_expression(e)
    for e in bad:
        with pytest.raises(pyoracle.OracleError):
            pyoracle.compile_expression(e)
    # validate_expression = compile + non-empty + no `in` (rules/rules.rs:55-77)
    pyoracle.validate_expression("a == 1")
    for e in ["", "x in [1]", "a ==", "[x in y]"]:
        with pytest.raises(pyoracle.OracleError):
            pyoracle.validate_expression(e)


def _exec(expr, req=None, lists=None):
    o = pyoracle.Oracle([("r", expr, [B])], lists, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
    return o.execute_rule(0, RequestBatch.from_requests([req or Request()]), 0)


def test_language_semantics_decisions():
    """One line per documented decision of DESIGN.md §3.3 (1 true, 0 false, 2 non-Bool, 3 error)."""
    R = Request(host="h.example", url="/a/b?x=1", path="/a/b", method="POST", user_agent="UA", ip="1.2.3.4", remote_port=8080, asn=64512, country="FR")
    cases = {
        # D3 member access == index
        'http_request["path"] == http_request.path': 1,
        # D4 equality across types is false, numeric Int/Float compare by value
        'http_request.path == 5': 0, 'client.remote_port == "8080"': 0, "1 == 1.0": 1, "client.remote_port == 8080.0": 1, "null == null": 1, "[1, 2] == [1, 2.0]": 1,
        'client.ip == "1.2.3.4"': 0,
        # D5 ordering only Int/Float/String
        '"a" < "b"': 1, "1 < 2.5": 1, 'client.remote_port < "9"': 3, "true < false": 3, "client.remote_port >= 8080": 1,
        # D6 && || left-to-right short circuit, Bool operands only, errors propagate when evaluated
        "true || http_request.nope": 1, "http_request.nope || true": 3, "false && http_request.nope": 0, "http_request.nope && false": 3,
        'http_request.path && true': 3, "false || 1": 3, "true || 1": 1,
        # D7 ! only on Bool
        "!true": 0, "!1": 3, "!!(1 == 1)": 1,
        # D8 arithmetic: checked Int, String concatenation
        "1 + 2 * 3 == 7": 1, "7 / 2 == 3": 1, "-7 % 3 == -1": 1, "1 / 0 == 1": 3, "9223372036854775807 + 1 > 0": 3, '"a" + "b" == "ab"': 1, "client.remote_port + 1 == 8081": 1,
        "1 + 1.5 == 2.5": 1, '1 + "a" == 1': 3,
        # D9 conditional evaluates only the chosen branch
        "true ? true : http_request.nope": 1, "false ? http_request.nope : true": 1, "1 ? true : false": 3,
        # D10 in
        "2 in [1, 2]": 1, '"k" in {"k": 1}': 1, "client.asn in [64512]": 1, '1 in "abc"': 3,
        # D11 functions and strict typing
        'http_request.path.contains("a/")': 1, "http_request.path.contains(1)": 3, 'http_request.path.starts_with("/a")': 1, 'http_request.path.ends_with("/b")': 1,
        "http_request.path.length() == 4": 1, "[1, 2, 3].length() == 3": 1, 'http_request.path.bogus()': 3, "length(http_request.path) == 4": 3, '"abc".contains("")': 1,
        '{"a": 1}.contains("a")': 1, 'http_request.contains("path")': 1, "[[1], [2]].contains([2])": 1,
        # D13 length counts bytes
        '"é".length() == 2': 1,
        # D14 matches
        'http_request.method.matches("^P(OS|U)T$")': 1, 'http_request.method.matches("(")': 3, 'http_request.url.matches("x=[0-9]+$")': 1,
        # non-Bool results never match (pingoo/rules.rs:47)
        "http_request.path": 2, "client.remote_port": 2, "null": 2, "[1]": 2,
        # unknown names are execution errors, not compile errors (rules/rules.rs:75 "validate variables TODO")
        "nope": 3, "client.nope": 3, 'lists["nope"]': 3, "http_request.path.nope": 3,
        # country and asn surface (pingoo/rules.rs:27-34)
        'client.country == "FR"': 1, "client.asn == 64512": 1, 'client.country.length() == 2': 1,
    }
    for expr, want in cases.items():
        assert _exec(expr, R) == want, expr


# ---- regex: the oracle's Pike VM against CPython's backtracking engine on the shared syntax -------------------
def _rand_regex(rng, depth=0):
    k = rng.randint(0, 12 if depth < 3 else 3)
    if k <= 1:
        return rng.choice(["a", "b", "c", "/", "\\.", "-", "ab", "\\n", "\\x41", "\\/"])
    if k == 2:
        return rng.choice(["[ab]", "[^a]", "[a-c/]", "\\w", "\\W", "\\d", "\\D", "\\s", "\\S", "[^/.]", "[\\w/]", "[a\\-c]", "[]a]", "[^]a]", "[a-]", "."])
    if k == 3:
        return rng.choice(["^", "$", "\\b", "\\B", "\\A"])
    if k == 4:
        return _rand_regex(rng, depth + 1) + _rand_regex(rng, depth + 1)
    if k == 5:
        return "(" + _rand_regex(rng, depth + 1) + "|" + _rand_regex(rng, depth + 1) + ")"
    if k == 6:
        return "(?:" + _rand_regex(rng, depth + 1) + ")" + rng.choice(["*", "+", "?", "{2}", "{1,3}", "{2,}", "*?", "+?", "??", "{0,2}?"])
    if k == 7:
        return _rand_regex(rng, depth + 1) + rng.choice(["a*", "b+", "/?", ".*", ".+", "[ab]{0,2}", "c{2}"])
    if k == 8:
        return "(?i:" + rng.choice(["A", "aB", "[A-B]c", "\\w"]) + ")" + _rand_regex(rng, depth + 1)
    if k == 9:
        return "(" + _rand_regex(rng, depth + 1) + ")" + _rand_regex(rng, depth + 1)
    if k == 10:
        return rng.choice(["a|", "|b", "(a|)", "()", "(?:)", "(?P<n>a)", "(?s:a.b)", "(?m:^a)", "(?m:b$)"])
    return _rand_regex(rng, depth + 1) + "|" + _rand_regex(rng, depth + 1)


def _to_python(pat: str) -> bytes:
    # Rust `$` (no multi-line) matches only at the very end; Python's also matches before a trailing '\n'
    out, i, in_class, ml = "", 0, False, 0
    while i < len(pat):
        ch = pat[i]
        if ch == "\\":
            out += pat[i:i + 2]
            i += 2
            continue
        if in_class:
            if ch == "]" and out[-1] not in "[^" :
                in_class = False
            out += ch
        elif ch == "[":
            in_class = True
            out += ch
            if pat[i + 1:i + 2] == "^":
                out += "^"
                i += 1
            if pat[i + 1:i + 2] == "]":
                out += "\\]"
                i += 1
        elif pat.startswith("(?m:", i):
            ml += 1
            out += "(?m:"
            i += 3
        elif ch == "$" and "(?m:" not in pat:
            out += "\\Z"
        else:
            out += ch
        i += 1
    return out.encode()


def test_regex_matches_cpython_re():
    rng = random.Random(1234)
    n_checked = 0
    for _ in range(1500):
        pat = _rand_regex(rng)
        if "(?m:" in pat and "$" in pat.replace("(?m:b$)", ""):
            continue  # mixed multi-line and plain `$` cannot be expressed with one translation rule
        try:
            pyre = re.compile(_to_python(pat))
        except re.error:
            continue
        for _ in range(12):
            hay = "".join(rng.choice("abc/.-A\n _1") for _ in range(rng.randint(0, 9))).encode()
            if hay == b"" and "\\B" in pat:
                continue  # CPython quirk (fixed in 3.14): \B never matches the empty string; the regex crate's does
            want = pyre.search(hay) is not None
            got = pyoracle.regex_is_match(pat, hay)
            assert got == want, (pat, hay, got, want)
            n_checked += 1
    assert n_checked > 10000


def _rand_uregex(rng, depth=0):
    """Patterns over the constructs whose Unicode semantics CPython's `re` (str mode) shares with the regex crate."""
    k = rng.randint(0, 11 if depth < 3 else 3)
    if k <= 1:
        return rng.choice(["a", "s", "k", "\u00e9", "\u03c3", "\u20ac", "a\u00e9", "\u0663", "\u00a0", "s\u00e9", "\U0001F600"])
    if k == 2:
        return rng.choice([".", "[^a]", "[^\u00e9]", "\\w", "\\W", "\\d", "\\D", "\\s", "\\S", "[\u00e0-\u00ff]", "[^\u0660-\u0669s]", "[a\u20ac]", "[\\w\u20ac]", "[\\d.]"])
    if k == 3:
        return rng.choice(["^", "$", "\\b", "\\B"])
    if k == 4:
        return _rand_uregex(rng, depth + 1) + _rand_uregex(rng, depth + 1)
    if k == 5:
        return "(" + _rand_uregex(rng, depth + 1) + "|" + _rand_uregex(rng, depth + 1) + ")"
    if k == 6:
        return "(?:" + _rand_uregex(rng, depth + 1) + ")" + rng.choice(["*", "+", "?", "{2}", "{1,3}", "{2,}"])
    if k == 7:
        return _rand_uregex(rng, depth + 1) + rng.choice([".*", ".+", ".{0,2}", "\\w*", "\\s+"])
    if k == 8:
        return "(?i:" + rng.choice(["S", "sK", "\u00c9", "\u03a3", "k\u00e9s", "\u017f"]) + ")" + _rand_uregex(rng, depth + 1)
    return _rand_uregex(rng, depth + 1) + "|" + _rand_uregex(rng, depth + 1)


def test_regex_unicode_matches_cpython_re_in_str_mode():
    """The oracle's Unicode half against an independent implementation: CPython `re` on the DECODED haystack folds s / U+017F and
    k / U+212A, treats \\d \\s \\w \\b as Unicode and `.` as one code point. The alphabet keeps to code points where the two \\w / \\s
    definitions agree (no marks, no No / Nl numbers, no U+001C-1F: CPython's isalnum / isspace differ from Alphabetic / White_Space there)."""
    rng = random.Random(4321)
    alphabet = ["a", "s", "S", "k", "K", "\u017f", "\u212a", "\u00e9", "\u00c9", "\u03c3", "\u03c2", "\u03a3", "\u20ac", "\u0663", "\u00a0", "\u2003", " ", ".", "_", "1", "\n", "\U0001F600"]
    n_checked = n_non_ascii_hits = 0
    for _ in range(2500):
        pat = _rand_uregex(rng)
        try:
            pyre = re.compile(_to_python(pat).decode())
        except re.error:
            continue
        for _ in range(10):
            hay = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 7)))
            if hay == "" and "\\B" in pat:
                continue  # (CPython quirk, see above)
            want = pyre.search(hay) is not None
            got = pyoracle.regex_is_match(pat, hay.encode())
            assert got == want, (pat, hay, got, want)
            n_checked += 1
            n_non_ascii_hits += want and not hay.isascii()
    assert n_checked > 15000 and n_non_ascii_hits > 3000


def test_unicode_tables_agree_with_cpython_unicodedata():
    """CPython carries its own copy of the Unicode Character Database (13.0.0): every general category, code point by code point, through
    the oracle's \\p{..}; \\d against str.isdecimal. The oracle's tables describe 14.0 since round 6 (perl's 13.0 + node's answer on what 14.0
    added, tools/merge_unicode_delta.py): compared on the code points 13.0 assigns, minus U+1734 whose category 14.0 changed (Mn -> Mc)."""
    import unicodedata

    assert unicodedata.unidata_version == "13.0.0"
    pts = [c for c in range(0x110000) if not 0xD800 <= c <= 0xDFFF and unicodedata.category(chr(c)) != "Cn" and c != 0x1734]
    text = "".join(chr(c) for c in pts)
    by_cat = {}
    for ch in text:
        by_cat.setdefault(unicodedata.category(ch), []).append(ch)
    for cat in sorted(by_cat):
        inside = "".join(by_cat[cat])
        assert pyoracle.regex_is_match("^\\p{%s}+$" % cat, inside.encode()), cat
        assert not pyoracle.regex_is_match("\\P{%s}" % cat, inside.encode()), cat
        outside = "".join("".join(v) for k, v in by_cat.items() if k != cat)
        assert not pyoracle.regex_is_match("\\p{%s}" % cat, outside.encode()), cat
    assert not pyoracle.regex_is_match("\\p{Cn}", text.encode())  # nothing 13.0 assigns is unassigned in 14.0
    assert pyoracle.regex_is_match("^\\p{Mc}$", "\u1734".encode()) and pyoracle.regex_is_match("^\\p{Vithkuqi}\\p{Lu}$", "\U00010570\U00010570".encode())  # 14.0
    digits = "".join(ch for ch in text if ch.isdecimal())
    assert pyoracle.regex_is_match("^\\d+$", digits.encode()) and not pyoracle.regex_is_match("\\d", "".join(ch for ch in text if not ch.isdecimal()).encode())


def test_unicode_tables_come_from_two_independent_sources(tmp_path):
    """oracle/unicode_data.inc and pingoo_amd/csrc/unicode_data.inc used to be the SAME file (one generator, perl's
    Unicode::UCD): a wrong script range, White_Space member or case-folding orbit was invisible to every product-vs-oracle test. Now the device
    compiler reads what node's ICU says (tools/gen_unicode_tables_node.js, Unicode 14.0: /\\p{..}/u and /c/iu over every scalar value) and the
    oracle what perl says on everything 13.0 assigns (+ node on the code points 14.0 added: tools/merge_unicode_delta.py). The generators
    are re-run here and compared: outside 13.0's unassigned space they may differ ONLY where the standard itself changed between the versions —
    U+1734 Mn -> Mc, U+16FE2/3 Common -> Han — and on the three case-folding orbits CaseFolding.txt gained in 15.1 (V8 folds them already)."""
    import shutil
    import subprocess
    import sys

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.join(root, "tools"))
    import merge_unicode_delta as M

    committed_oracle, committed_product = os.path.join(root, "oracle", "unicode_data.inc"), os.path.join(root, "pingoo_amd", "csrc", "unicode_data.inc")
    assert open(committed_product).readline().startswith("// GENERATED by tools/gen_unicode_tables_node.js") and "merge_unicode_delta.py" in open(committed_oracle).readline()
    ot, of = M.parse(committed_oracle)
    pt, pf = M.parse(committed_product)
    do, dp = {M.key(t): M.to_set(t[2]) for t in ot}, {M.key(t): M.to_set(t[2]) for t in pt}
    assert set(do) == set(dp) and len(do) > 200
    assert all(do[k] == dp[k] for k in do) and set(of) == set(pf)  # the two committed files describe the same sets (from different sources)
    if not (shutil.which("perl") and shutil.which("node")) or subprocess.run(["perl", "-MUnicode::UCD", "-e", "1"]).returncode != 0:
        pytest.skip("perl (Unicode::UCD) and node are needed to re-run the two generators")
    perl_out, node_out = tmp_path / "perl.inc", tmp_path / "node.inc"
    perl_out.write_bytes(subprocess.run(["perl", os.path.join(root, "tools", "gen_unicode_tables.pl")], check=True, stdout=subprocess.PIPE).stdout)
    node_out.write_bytes(subprocess.run(["node", "--max-old-space-size=4096", os.path.join(root, "tools", "gen_unicode_tables_node.js")], check=True, stdout=subprocess.PIPE).stdout)
    assert node_out.read_bytes() == open(committed_product, "rb").read()  # the product's file IS node's output
    a, fa = M.parse(str(perl_out))
    b, fb = M.parse(str(node_out))
    cn13 = M.unassigned(a)
    assert 830 <= len(cn13 - M.unassigned(b)) <= 850  # 14.0 added 838 characters
    da, db = {M.key(t): M.to_set(t[2]) for t in a}, {M.key(t): M.to_set(t[2]) for t in b}
    diff = set()
    for k in set(da) | set(db):
        diff |= (da.get(k, set()) ^ db.get(k, set())) - cn13
    assert diff == M.CHANGED_IN_14, sorted(hex(x) for x in diff)
    assert {p for p in set(fa) ^ set(fb) if p[0] not in cn13 and p[1] not in cn13} == M.FOLDS_SINCE_15_1
    # ... and the oracle's committed file is the merge of the two
    merged = subprocess.run([sys.executable, os.path.join(root, "tools", "merge_unicode_delta.py"), str(perl_out), str(node_out)], check=True, stdout=subprocess.PIPE).stdout
    assert merged == open(committed_oracle, "rb").read()


def test_regex_syntax_errors_and_unsupported():
    for pat in ["(", ")", "a{2", "a{,3}", "*a", "a**b{", "[a", "\\", "\\q", "(?z)", "(?P<>a)", "a{3,2}", "[b-a]", "(?=a)", "(?<!a)", "\\1", "\\p{Bogus}", "\\p{age=6.0}", "\\p{", "[[:bogus:]]", "[a&&b]", "\\xZZ",
                "\\x{110000}", "\\x{D800}", "(?R)a", "(?-u:.)", "(?-u:\\W)", "(?-u:[^a])", "(?-u:\\xFF)", "(?-u:\\pL)", "\\b{start}a", "\\<a"]:
        with pytest.raises(pyoracle.OracleError):
            pyoracle.regex_is_match(pat, b"a")
    # (?-u): only the two-digit \xHH is a byte; \x{..} \uHHHH \UHHHHHHHH stay scalar values (regex-syntax Literal::byte(): ADVICE r5)
    assert pyoracle.regex_is_match("(?-u:\\x{e9}\\u00e9\\U000000e9)$", "\u00e9\u00e9\u00e9".encode()) and not pyoracle.regex_is_match("(?-u:\\x{e9})", b"\xe9")
    for pat in ["(?-u:[\\x{e9}])", "(?-u:\\xe9)"]:
        with pytest.raises(pyoracle.OracleError):
            pyoracle.regex_is_match(pat, b"a")
    assert pyoracle.regex_is_match("[[:alpha:]]+[[:digit:]]", b"..ab1")
    assert pyoracle.regex_is_match("^\\p{L}+\\p{N}$", b"ab1") and not pyoracle.regex_is_match("^\\P{L}$", b"a")  # general categories, over ASCII
    assert pyoracle.regex_is_match("(?i)union\\s+select", b"x UnIoN \t SELECT y")
    assert not pyoracle.regex_is_match("^$", b"a") and pyoracle.regex_is_match("^$", b"") and pyoracle.regex_is_match("a*", b"")
    assert pyoracle.regex_is_match("a$", b"a") and not pyoracle.regex_is_match("a$", b"a\n")  # Rust `$` is not Perl's
    assert pyoracle.regex_is_match("(?m)a$", b"a\nb") and pyoracle.regex_is_match("(?m)^b", b"a\nb")
    assert not pyoracle.regex_is_match("a.b", b"a\nb") and pyoracle.regex_is_match("(?s)a.b", b"a\nb")


def test_regex_unicode_semantics_of_the_regex_crate():
    """regex 1.12.2 (Cargo.lock:1694-1700) is Unicode-aware by default and http 1.3.1 (Cargo.lock:824-826) admits UTF-8 in path and
    query, which reach `bel` as Rust str (pingoo/rules.rs:16-25): VERDICT r4 Missing #2. Hand-derived from the crate's documented
    semantics (regex-syntax's translation: `.` / negated classes = one scalar value, \\d = Nd, \\s = White_Space, \\w = Alphabetic + M + Nd +
    Pc + Join_Control, (?i) = simple case folding, \\b Unicode-aware, (?-u) = ASCII)."""
    M = lambda pat, hay: pyoracle.regex_is_match(pat, hay.encode() if isinstance(hay, str) else hay)
    # the fail-open cases VERDICT names
    assert M("(?i)union\\s+select", "q=union\u00a0select") and M("(?i)union\\s+select", "q=UNION\u2003\u3000SELECT")
    assert M("(?i)select", "\u017felect") and M("(?i)nikto", "ni\u212ato") and not M("select", "\u017felect")
    assert M("^\\w+$", "caf\u00e9") and M("^\\w+$", "\u4f60\u597d_1") and not M("^\\w+$", "a\u20acb")
    # one scalar value per `.` / negated class
    assert M("^a.b$", "a\u00e9b") and M("^a.b$", "a\U0001F600b") and not M("^a..b$", "a\u00e9b") and M("^a[^x]b$", "a\u20acb")
    assert M("^.{3}$", "\u00e9\u20ac\U0001F600") and not M("^.{3}$", "\u00e9\u20ac")
    # \\d is Nd, \\s is White_Space, \\D \\S \\W their complements over scalar values
    assert M("^\\d+$", "\u0663\u0967") and not M("^\\d$", "\u00b2") and M("^\\D$", "\u00b2") and M("^\\S$", "\u00e9") and not M("^\\S$", "\u2028")
    assert M("^\\W$", "\u20ac") and not M("^\\W$", "\u00e9") and not M("\\s", "\x1c") and M("\\s", "\x85".encode("latin1").decode("latin1"))
    # Unicode word boundaries
    assert not M("\\bselect", "\u00e9select") and M("\\bselect", "\u20acselect") and M("(?-u:\\b)select", "\u00e9select")
    assert not M("select\\b", "select\u00e9") and M("select\\B", "select\u00e9") and M("select\\b", "select\u2003x")
    # (?-u): ASCII classes, ASCII folding
    assert not M("(?-u:\\w)$", "\u00e9") and not M("(?i-u)select", "\u017felect") and M("(?i-u)select", "SELECT") and not M("(?-u:\\s)", "\u00a0")
    # classes: ranges over code points, folding of the whole class, POSIX classes stay ASCII (but fold: K is in the orbit of k)
    assert M("^[\u03b1-\u03c9]+$", "\u03b1\u03c9") and M("(?i)^[\u03b1-\u03c9]+$", "\u0391\u03a9") and M("(?i)^[[:lower:]]$", "\u212a") and not M("^[[:lower:]]$", "\u212a")
    assert M("^[[:^alpha:]]$", "\u00e9") and M("^\\x{e9}\\u00e9\\u{e9}\\U000000e9$", "\u00e9" * 4) and M("^\\xe9$", "\u00e9")
    # \\p{..}: categories, scripts, binary properties; folded under (?i), then negated
    assert M("^\\p{Greek}+$", "\u03b1\u03b2") and M("^\\p{sc=Cyrillic}$", "\u0436") and M("^\\p{Lu}\\p{Ll}$", "\u00c9\u00e9") and M("(?i)^\\p{Lu}$", "\u00e9")
    assert M("^\\P{L}$", "\u20ac") and not M("^\\P{L}$", "\u00e9") and M("^\\p{gc=Nd}\\p{Alphabetic}\\p{White_Space}$", "\u0663\u00e9\u3000") and M("^\\pN$", "\u00b2")
    # (?x): whitespace and comments are syntax
    assert M("(?x) union \\s+ select  # comment", "union select") and M("(?x)a\\ b [ c d ]", "a bd") and not M("(?x)a b", "a b") and M("(?x)a{1, 2}$", "aa")
    # ill-formed UTF-8 (unreachable through the reference's str): a unit no class matches; \\b and \\B are both false next to it (D17)
    assert not M("a.b", b"a\xffb") and not M("a[^x]b", b"a\xffb") and M("a", b"\xffa\xff") and not M("\\ba", b"\xffa") and not M("\\Ba", b"\xffa")
    assert not M("^a.b$", b"a\xc3b") and not M("^.$", b"\xed\xa0\x80") and not M("^.$", b"\xc0\x80") and M("^.$", b"\xf4\x8f\xbf\xbf") and not M("^.$", b"\xf4\x90\x80\x80")


# ---- ip parsing / containment against the stdlib -------------------------------------------------------------------
def test_ip_parse_and_cidr_containment_match_ipaddress():
    rng = random.Random(99)
    texts = ["1.2.3.4", "01.2.3.4", "1.2.3", "1.2.3.4.5", "256.1.1.1", "1.2.3.04", "", " 1.2.3.4", "::", "::1", "1::", "2001:db8::1", "2001:db8:0:0:0:0:0:1", "1:2:3:4:5:6:7:8",
             "1:2:3:4:5:6:7:8:9", "::ffff:1.2.3.4", "1:2:3:4:5:6:1.2.3.4", "1::2::3", ":1", "1:", "12345::", "g::", "::1.2.3", "1:2:3:4:5:6:7::", "::2:3:4:5:6:7:8", "1:2:3:4:5:6:7::8"]
    for t in texts:
        fam, raw = pyoracle.parse_ip(t)
        try:
            a = ipaddress.ip_address(t)
            assert fam == a.version, t
            assert raw[: len(a.packed)] == a.packed, t
        except ValueError:
            assert fam == 0, t
    for _ in range(4000):
        if rng.random() < 0.6:
            addr = ipaddress.IPv4Address(rng.getrandbits(32))
            plen = rng.randint(0, 32)
        else:
            addr = ipaddress.IPv6Address(rng.getrandbits(128))
            plen = rng.randint(0, 128)
        net = ipaddress.ip_network(f"{addr}/{plen}", strict=False)
        # ipnetwork keeps host bits in the text; containment masks both sides
        probe_int = int(net.network_address) + rng.getrandbits(net.max_prefixlen - plen) if (plen < net.max_prefixlen and rng.random() < 0.5) else rng.getrandbits(net.max_prefixlen)
        probe = type(addr)(probe_int)
        raw = probe.packed + b"\0" * (16 - len(probe.packed))
        assert pyoracle.ipnet_contains(f"{addr}/{plen}", raw, probe.version == 6) == (probe in net)
        other = ipaddress.IPv6Address(1) if addr.version == 4 else ipaddress.IPv4Address(1)
        assert not pyoracle.ipnet_contains(f"{addr}/{plen}", other.packed + b"\0" * (16 - len(other.packed)), other.version == 6)
    assert pyoracle.ipnet_contains("10.0.0.0/255.0.0.0", bytes([10, 9, 8, 7]) + b"\0" * 12, False)
    for bad in ["10.0.0.0/33", "10.0.0.0/255.0.255.0", "::/129", "1.2.3.4/", "1.2.3.4/a", "x/8", "1.2.3.4/-1"]:
        with pytest.raises(pyoracle.OracleError):
            pyoracle.ipnet_contains(bad, b"\0" * 16, False)


def test_geoip_longest_prefix_and_defaults():
    rows = [("1.0.0.0/8", 10, "AU"), ("1.2.0.0/16", 20, "CN"), ("1.2.3.0/24", 30, "FR"), ("1.2.3.0/24", 31, "DE"), ("5.0.0.0/8", 40, "q1"), ("5.5.0.0/16", 50, "US"),
            ("127.0.0.0/8", 60, "US"), ("224.0.0.0/4", 61, "US"), ("2001:db8::/32", 70, "JP"), ("::/0", 71, "BR"), ("0.0.0.0/0", 5, "ZZ")]
    o = pyoracle.Oracle([("r", None, [B])], None, geoip_entries(rows))

    def look(ip):
        a = ipaddress.ip_address(ip)
        return o.geoip_lookup(a.packed + b"\0" * (16 - len(a.packed)), a.version == 6)

    assert look("1.9.9.9") == (10, b"AU") and look("1.2.9.9") == (20, b"CN")
    assert look("1.2.3.4") == (31, b"DE")  # the later duplicate wins
    assert look("5.1.1.1") == (0, b"XX")   # record with an invalid country fails to decode -> default (geoip.rs:128-142, http_listener.rs:148-153)
    assert look("5.5.1.1") == (50, b"US")
    assert look("127.0.0.1") == (0, b"XX") and look("224.0.0.9") == (0, b"XX")  # loopback / multicast are never looked up (geoip.rs:74-76)
    assert look("::1") == (0, b"XX") and look("ff02::1") == (0, b"XX")
    assert look("2001:db8::5") == (70, b"JP") and look("2001:db9::5") == (71, b"BR") and look("9.9.9.9") == (5, b"ZZ")
    no_db = pyoracle.Oracle([("r", None, [B])])
    assert no_db.geoip_lookup(bytes([1, 2, 3, 4]) + b"\0" * 12, False) == (0, b"XX")


def test_list_loading_errors_and_trimming():
    o = pyoracle.Oracle([("r", 'lists["a"].contains(client.asn) && lists["s"].contains(http_request.method) && lists["n"].contains(client.ip)', [B])],
                        {"a": (_abi.LIST_INT, [" 7 ", "-3", "+5"]), "s": (_abi.LIST_STRING, ["  GET\t", ""]), "n": (_abi.LIST_IP, [" 9.9.9.9 "])})
    b = RequestBatch.from_requests([Request(method="GET", ip="9.9.9.9", asn=7, country="FR"), Request(method="GET ", ip="9.9.9.9", asn=7, country="FR")])
    assert [int(v["action"]) for v in o.evaluate(b)] == [1, 0]
    for lists in [{"a": (_abi.LIST_INT, ["x"])}, {"a": (_abi.LIST_INT, ["1.5"])}, {"a": (_abi.LIST_INT, [""])}, {"a": (_abi.LIST_INT, ["9223372036854775808"])},
                  {"n": (_abi.LIST_IP, ["1.2.3"])}, {"n": (_abi.LIST_IP, ["1.2.3.4/40"])}]:
        with pytest.raises(pyoracle.OracleError) as ei:
            pyoracle.Oracle([("r", None, [B])], lists)
        assert ei.value.code == _abi.E_LIST


def test_regex_rule_sets_of_the_synthetic_configs_match_cpython_re():
    """tools/regex_crosscheck.py at reduced size: every `matches` pattern of BASELINE configs[2] / configs[4] against the field values
    of benign and adversarial requests, oracle regex engine vs CPython `re` (the full run is the script's default)."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("regex_crosscheck", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "regex_crosscheck.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n_pats, checked, bad = mod.crosscheck(3, 150, verbose=False)
    assert n_pats >= 150 and checked > 50_000 and bad == 0


def test_regex_unicode_matches_node_u_mode():
    """VERDICT r5 #5c: ECMAScript RegExp in `u` mode (node: V8 + ICU) as a THIRD implementation of the scalar-value regex semantics, on the
    constructs whose meaning coincides with the regex crate's: `.` (s flag) and negated classes take one scalar value, \\p{..} general
    categories and scripts, explicit ranges beyond ASCII, quantifiers over multi-byte scalars, and (?i) = simple case folding on literals and
    positive ranges (U+017F ~ s, U+212A ~ k). 6 000 generated (pattern, haystack) pairs; tools/regex_node_crosscheck.js evaluates them."""
    import json
    import shutil
    import subprocess

    if not shutil.which("node"):
        pytest.skip("node is not installed")
    rng = random.Random(20260930)
    alpha = ["a", "b", "k", "s", "S", "K", "z", "0", "7", "_", "-", " ", "/", "é", "É", "ſ", "K", "€", "\U0001F600", "α", "Ω", "ж", "Ж", "٣",
             " ", "你", "́", "᜴", "\U00010570"]
    classes_plain = ["[a-k]", "[α-ω]", "[а-я]", "[aé€]", "[0-9٠-٩]"]
    classes_unicode = ["\\p{L}", "\\p{Lu}", "\\p{Ll}", "\\P{L}", "\\p{Nd}", "\\p{Greek}", "\\p{Cyrillic}", "\\p{Mn}", "\\p{Mc}", "\\p{Vithkuqi}", "[^a-c]", "[^\\p{L}]", "[\\p{L}\\p{Nd}_]", "[^é]", "."]

    def atom(ci):
        r = rng.random()
        if r < 0.5:
            ch = rng.choice(alpha)
            return "\\" + ch if ch in "-/ " and rng.random() < 0.2 and ch != " " else ch
        if r < 0.7 or ci:
            return rng.choice(classes_plain)
        return rng.choice(classes_unicode)

    def piece(ci, depth=0):
        a = atom(ci) if depth > 1 or rng.random() < 0.8 else "(?:" + expr(ci, depth + 1) + ")"
        return a + rng.choice(["", "", "", "*", "+", "?", "{2}", "{1,3}"])

    def expr(ci, depth=0):
        alts = ["".join(piece(ci, depth) for _ in range(rng.randint(1, 4))) for _ in range(rng.choice([1, 1, 1, 2]))]
        return "|".join(alts)

    cases = []
    for _ in range(6000):
        ci = rng.random() < 0.3
        pat = expr(ci)
        if rng.random() < 0.3:
            pat = "^" + pat
        if rng.random() < 0.3:
            pat = pat + "$"
        hay = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 8)))
        cases.append((pat, ci, hay))
    js_in = [[p.replace("\\p{Greek}", "\\p{Script=Greek}").replace("\\p{Cyrillic}", "\\p{Script=Cyrillic}").replace("\\p{Vithkuqi}", "\\p{Script=Vithkuqi}").replace("\\-", "-").replace("\\/", "/"),
              "su" + ("i" if ci else ""), h] for p, ci, h in cases]
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "regex_node_crosscheck.js")
    res = json.loads(subprocess.run(["node", tool], input=json.dumps(js_in).encode(), stdout=subprocess.PIPE, check=True).stdout)
    n_true = n_multi = 0
    for (pat, ci, hay), js in zip(cases, res):
        assert not isinstance(js, str), (pat, js)
        got = pyoracle.regex_is_match("(?s" + ("i" if ci else "") + ")" + pat, hay.encode())
        assert got == js, (pat, ci, hay, got, js)
        n_true += js
        n_multi += js and not hay.isascii()
    assert n_true > 900 and n_multi > 500, (n_true, n_multi)
