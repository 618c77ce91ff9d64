"""Shared test helpers: golden-vector loading, a random expression / request fuzzer, verdict comparison."""
from __future__ import annotations

import json
import os
import random

import numpy as np

from pingoo_amd import Request, RequestBatch, _abi

HERE = os.path.dirname(os.path.abspath(__file__))
B, CAP = _abi.RULE_ACTION_BLOCK, _abi.RULE_ACTION_CAPTCHA


def load_kat():
    with open(os.path.join(HERE, "golden", "kat.json")) as f:
        return json.load(f)


def kat_case_inputs(c):
    rules = [(r[0], r[1], r[2]) for r in c["rules"]]
    lists = {k: (v[0], v[1]) for k, v in c.get("lists", {}).items()}
    batch = RequestBatch.from_requests([Request(**r) for r in c["requests"]])
    expect = np.array([(a, r) for a, r in c["expect"]], dtype=np.int64)
    return rules, lists, batch, expect


def assert_verdicts_equal(got, want, batch=None, what=""):
    bad = np.nonzero((got["action"] != want["action"]) | (got["rule_idx"] != want["rule_idx"]))[0]
    if len(bad):
        i = int(bad[0])
        ctx = ""
        if batch is not None:
            ctx = " fields=" + repr([batch.field_bytes(f, i) for f in range(5)]) + f" ip={bytes(batch.ip[i]).hex()} v6={batch.ip_is_v6[i]} port={batch.port[i]} flags={batch.flags[i]}"
        raise AssertionError(f"{what}: {len(bad)} of {len(got)} verdicts differ; first at {i}: got ({got[i]['action']}, {got[i]['rule_idx']}) "
                             f"want ({want[i]['action']}, {want[i]['rule_idx']}){ctx}")


# ---------------------------------------------------------------------------------------------------------
# fuzzer: small alphabets so that predicates actually fire
# ---------------------------------------------------------------------------------------------------------
ALPHA = "ab/."
# url / path / host values with UTF-8 in them (http 1.3.1 admits it in path and query, the regex crate matches scalar values): two- and
# three-byte sequences next to the ASCII alphabet, members of \w (e-acute), of \s (U+00A0), of neither (the euro sign), and the two
# code points whose simple case folding reaches ASCII (U+017F ~ s, U+212A ~ k)
UALPHA = "ab/.sk\u00e9\u20ac\u00a0\u017f\u212a"
FIELDS = ["host", "url", "path", "method", "user_agent"]
COUNTRIES = ["XX", "FR", "US", "CN", "DE", "AA", "ZZ"]


def rstr(rng: random.Random, lo=0, hi=6, alpha=ALPHA):
    return "".join(rng.choice(alpha) for _ in range(rng.randint(lo, hi)))


def q(s: str) -> str:
    return '"' + s.replace("\\", "\\\\").replace('"', '\\"') + '"'


def rregex(rng: random.Random, depth=0) -> str:
    """Random pattern over the syntax both the oracle and the device compiler support."""
    k = rng.randint(0, 13 if depth < 3 else 4)
    if k <= 1:
        return rng.choice(["a", "b", "/", "\\.", "ab", "ba", "a/b", ".", "\u00e9", "a\u20ac", "s"])
    if k == 2:
        if rng.random() < 0.2:  # Unicode general categories
            return rng.choice(["\\p{L}", "\\pL", "\\P{L}", "\\p{Lu}", "\\p{Ll}", "\\p{N}", "\\p{Nd}", "\\P{N}", "\\p{P}", "\\p{^P}", "[\\p{L}/]", "[^\\p{N}a]", "\\p{S}", "\\p{Zs}", "[\\P{Lu}b]",
                               "[^\u00e9]", "[\u00e0-\u00ff]", "(?-u:\\w)", "(?-u:\\b)"])
        return rng.choice(["[ab]", "[^a]", "[a-b/]", "\\w", "\\W", "\\d", "\\s", "[[:alpha:]]", "[^/.]", "\\S"])
    if k == 3:
        return rng.choice(["^", "$", "\\b", "\\B", "\\A", "\\z"]) if rng.random() < 0.5 else "a"
    if k == 4:
        return rregex(rng, depth + 1) + rregex(rng, depth + 1)
    if k == 5:
        return "(" + rregex(rng, depth + 1) + "|" + rregex(rng, depth + 1) + ")"
    if k == 6:
        return "(?:" + rregex(rng, depth + 1) + ")" + rng.choice(["*", "+", "?", "{2}", "{1,3}", "{2,}", "*?", "+?"])
    if k == 7:
        return rregex(rng, depth + 1) + rng.choice(["a*", "b+", "/?", ".*", ".+", "[ab]{0,2}"])
    if k == 8:
        return "^" + rregex(rng, depth + 1)
    if k == 9:
        return rregex(rng, depth + 1) + "$"
    if k == 10:
        return "(?i)" + rng.choice(["A", "aB", "[A-B]/", "S", "Ks", "\u00c9", "[r-t]"]) + rregex(rng, depth + 1)
    if k == 11:
        return "(" + rregex(rng, depth + 1) + ")" + rregex(rng, depth + 1)
    if k == 12:
        return rng.choice(["(?m)^a", "(?m)b$", "(?s)a.b", "a|", "|b", "(a|)", "()", "(?i:a)b"])
    return rregex(rng, depth + 1) + "|" + rregex(rng, depth + 1)


def rpred(rng: random.Random, lists) -> str:
    """One atomic predicate (mostly well-typed, sometimes deliberately not)."""
    k = rng.randint(0, 30)
    f = "http_request." + rng.choice(FIELDS)
    if k <= 4:
        return f"{f}.{rng.choice(['contains', 'starts_with', 'ends_with'])}({q(rstr(rng, 0, 3))})"
    if k <= 6:
        return f"{f} {rng.choice(['==', '!='])} {q(rstr(rng, 0, 3))}"
    if k <= 9:
        return f"{f}.matches({q(rregex(rng))})"
    if k == 10:
        return f"{f}.length() {rng.choice(['==', '!=', '<', '<=', '>', '>='])} {rng.randint(-1, 6)}"
    if k == 11:
        return f"client.remote_port {rng.choice(['==', '!=', '<', '<=', '>', '>='])} {rng.choice([0, 1, 2, 3, 80, 443, 65535, 70000, -5, 2.5, 3.0])}"
    if k == 12:
        return f"client.asn {rng.choice(['==', '<', '>='])} {rng.choice([0, 1, 2, 3, 4294967295, 64512])}"
    if k == 13:
        items = ", ".join(q(rstr(rng, 0, 3)) for _ in range(rng.randint(0, 4)))
        return f"[{items}].contains({f})"
    if k == 14:
        items = ", ".join(str(rng.randint(0, 5)) for _ in range(rng.randint(0, 4)))
        v = rng.choice(["client.asn", "client.remote_port", f + ".length()"])
        return rng.choice([f"[{items}].contains({v})", f"{v} in [{items}]"])
    if k == 15:
        return f"client.country {rng.choice(['==', '!='])} {q(rng.choice(COUNTRIES))}"
    if k == 16:
        items = ", ".join(q(rng.choice(COUNTRIES)) for _ in range(rng.randint(1, 3)))
        return f"[{items}].contains(client.country)"
    if k == 17:
        return f"client.country.{rng.choice(['starts_with', 'ends_with', 'contains'])}({q(rng.choice(['X', 'F', 'R', 'XX', '', 'U']))})"
    if k == 18 and lists:
        ip_lists = [n for n, (t, _) in lists.items() if t == _abi.LIST_IP]
        if ip_lists:
            n = rng.choice(ip_lists)
            return rng.choice([f"lists[{q(n)}].contains(client.ip)", f"client.ip in lists.{n}"])
    if k == 19 and lists:
        s_lists = [n for n, (t, _) in lists.items() if t == _abi.LIST_STRING]
        if s_lists:
            return f"lists[{q(rng.choice(s_lists))}].contains({f})"
    if k == 20 and lists:
        i_lists = [n for n, (t, _) in lists.items() if t == _abi.LIST_INT]
        if i_lists:
            return f"lists[{q(rng.choice(i_lists))}].contains(client.asn)"
    if k == 21:
        return rng.choice(["true", "false", "1 == 1", '"a" < "b"', "1 + 1 == 2", "[1, 2].contains(2)", '"abc".contains("b")', "2 > 3.5", "1 / 0 == 1", '"x".length() == 1'])
    if k == 22:  # statically erroring or ill-typed
        return rng.choice(["http_request.nope == 1", "client.nope", 'lists["missing"].contains(client.ip)', f"{f} && true", "!client.remote_port", "undefined_fn(1)",
                           f'{f}.contains(1)', f"{f}.bogus()", 'client.ip == "1.2.3.4"', "client.remote_port == \"80\"", "1 < \"a\"", f'{f}.matches("(")', "null == null"])
    if k == 23:
        return f"{q(rstr(rng, 1, 5))}.{rng.choice(['contains', 'starts_with', 'ends_with'])}({f})"
    if k == 24:
        return f"{f}.length() {rng.choice(['<', '>='])} {rng.randint(0, 5)}.5"
    if k == 25:
        return f"{f} in [{q(rstr(rng, 0, 2))}, {q(rstr(rng, 0, 3))}]"
    if k == 26:
        return f'http_request["{rng.choice(FIELDS)}"] == {q(rstr(rng, 0, 2))}'
    if k == 27:
        return f'"path" in http_request && {f}.contains({q(rstr(rng, 1, 2))})'
    return f"{f}.contains({q(rstr(rng, 1, 2))})"


def rexpr(rng: random.Random, lists, depth=0) -> str:
    k = rng.randint(0, 9 if depth < 3 else 2)
    if k <= 2:
        return rpred(rng, lists)
    if k == 3:
        return "!" + ("(" + rexpr(rng, lists, depth + 1) + ")")
    if k in (4, 5):
        return "(" + rexpr(rng, lists, depth + 1) + rng.choice([" && ", " || "]) + rexpr(rng, lists, depth + 1) + ")"
    if k == 6:
        return rexpr(rng, lists, depth + 1) + rng.choice([" && ", " || "]) + rexpr(rng, lists, depth + 1)
    if k == 7:
        return "(" + rexpr(rng, lists, depth + 1) + " ? " + rexpr(rng, lists, depth + 1) + " : " + rexpr(rng, lists, depth + 1) + ")"
    if k == 8:
        return "(" + rexpr(rng, lists, depth + 1) + rng.choice([" == ", " != "]) + rexpr(rng, lists, depth + 1) + ")"
    return rpred(rng, lists)


def fuzz_lists(rng: random.Random):
    nets = []
    for _ in range(rng.randint(1, 12)):
        if rng.random() < 0.75:
            ln = rng.choice([8, 16, 24, 30, 32, 0, 12, 31])
            nets.append(f"{rng.randint(1, 3)}.{rng.randint(0, 3)}.{rng.randint(0, 3)}.{rng.randint(0, 255)}/{ln}")
        else:
            nets.append(f"2001:db8:{rng.randint(0, 3):x}::{rng.randint(0, 3):x}/{rng.choice([32, 48, 64, 128, 127, 0, 3])}")
    nets2 = [f"{rng.randint(1, 3)}.{rng.randint(0, 3)}.0.0/{rng.choice([14, 15, 16, 23])}" for _ in range(rng.randint(0, 5))] + ["1.1.1.1", " 2.2.2.2 "]
    return {
        "nets": (_abi.LIST_IP, nets),
        "nets2": (_abi.LIST_IP, nets2),
        "words": (_abi.LIST_STRING, [rstr(rng, 0, 3) for _ in range(rng.randint(0, 6))]),
        "asns": (_abi.LIST_INT, [str(rng.randint(0, 5)) for _ in range(rng.randint(0, 5))] + [" 64512 "]),
    }


def fuzz_requests(rng: random.Random, n: int, with_geo: bool):
    reqs = []
    for _ in range(n):
        alpha = UALPHA if rng.random() < 0.3 else ALPHA
        path = rstr(rng, 0, 8, alpha)
        if rng.random() < 0.1:
            path = "/__pingoo/captcha" + path
        ua = rstr(rng, 0, 8, "abM/ ") if rng.random() < 0.9 else rng.choice(["", "x" * 255, "x" * 256, "x" * 290])
        if rng.random() < 0.75:
            ip = f"{rng.randint(1, 3)}.{rng.randint(0, 3)}.{rng.randint(0, 3)}.{rng.randint(0, 255)}"
        else:
            ip = rng.choice([f"2001:db8:{rng.randint(0, 3):x}::{rng.randint(0, 3):x}", "::1", "ff02::1", "127.0.0.1", "224.1.2.3", "1.1.1.1", "2.2.2.2"])
        kw = {}
        if with_geo:
            kw = dict(asn=rng.choice([0, 1, 2, 3, 64512, 4294967295]), country=rng.choice(COUNTRIES))
        reqs.append(Request(host=rstr(rng, 0, 6, alpha), url=path + rstr(rng, 0, 4, alpha), path=path, method=rng.choice(["GET", "POST", "a", ""]), user_agent=ua, ip=ip,
                            remote_port=rng.choice([0, 1, 2, 3, 80, 443, 65535, rng.randint(0, 65535)]), captcha_verified=rng.random() < 0.3, **kw))
    return reqs


def fuzz_geoip(rng: random.Random):
    from pingoo_amd import geoip_entries

    rows = []
    for _ in range(rng.randint(1, 10)):
        if rng.random() < 0.7:
            rows.append((f"{rng.randint(1, 3)}.{rng.randint(0, 3)}.0.0/{rng.choice([8, 14, 16, 24])}", rng.choice([0, 1, 2, 3, 64512]), rng.choice(COUNTRIES + ["xx", "F1"])))
        else:
            rows.append((f"2001:db8:{rng.randint(0, 3):x}::/{rng.choice([32, 48, 64])}", rng.choice([1, 2, 3]), rng.choice(COUNTRIES)))
    rows.append(("127.0.0.0/8", 3, "US"))  # must be ignored: loopback is never looked up (geoip.rs:74-76)
    rows.append(("::/0", 2, "DE"))
    return geoip_entries(rows)


def fuzz_actions(rng: random.Random):
    return rng.choice([[B], [CAP], [CAP, B], [B, CAP], [], [CAP, CAP]])


# ---------------------------------------------------------------------------------------------------------
# literal-heavy rule sets: every pattern has a literal factor of two or more bytes, so the compiler puts the passes behind
# the bigram prefilter; requests are built from the same tokens (whole, truncated, case-flipped, glued across
# field boundaries) so that both matches and near misses are frequent
# ---------------------------------------------------------------------------------------------------------
TOKENS = ["/.env", "../", "<script", "union select", "/wp-admin", "cmd.exe", "ab", "Mozilla/", "curl/", "bot", "x9k2", "/api/v1", "=%27", "admin", "select", "q7",
          ".php", "passwd", "etc", "AbCd", "zz", "/a/b", "hello-world", "0x41", "id=", "__", "a.b"]


def lit_token(rng: random.Random) -> str:
    t = rng.choice(TOKENS)
    k = rng.random()
    if k < 0.15 and len(t) > 3:
        t = t[:rng.randint(2, len(t) - 1)]
    elif k < 0.25:
        t = t + rng.choice(TOKENS)
    return t


HEADER_NAMES = ["x-a", "cookie", "x-b"]


def lit_pred(rng: random.Random) -> str:
    f = "http_request." + rng.choice(["host", "url", "path", "user_agent", "url", "path", 'headers["x-a"]', 'headers["cookie"]', "headers.referer"])
    t = lit_token(rng)
    k = rng.randint(0, 13)
    if k == 12:
        return f"{f}.length() {rng.choice(['>', '<=', '=='])} {rng.randint(0, 12)}"
    if k == 13:
        return rng.choice(['"x-a" in http_request.headers', 'http_request.headers.contains("cookie")', '"headers" in http_request', f'{f} == ""'])
    if k <= 2:
        return f"{f}.contains({q(t)})"
    if k == 3:
        return f"{f}.starts_with({q(t)})"
    if k == 4:
        return f"{f}.ends_with({q(t)})"
    if k == 5:
        return f"{f} == {q(t)}"
    esc = "".join("\\" + c if c in ".^$*+?()[]{}|\\/" else c for c in t)
    if k == 6:
        return f"{f}.matches({q('(?i)' + esc)})"
    if k == 7:
        return f"{f}.matches({q(esc + '[0-9]+' + ''.join(chr(92) + c if c in '.^$*+?()[]{}|/' else c for c in lit_token(rng)))})"
    if k == 8:
        return f"{f}.matches({q(esc + '.*' + ''.join(chr(92) + c if c in '.^$*+?()[]{}|/' else c for c in lit_token(rng)))})"
    if k == 9:
        return f"{f}.matches({q('^' + esc + '(x|y|' + esc + ')$')})"
    if k == 10:
        return f"{f}.matches({q('(' + esc + '|' + ''.join(chr(92) + c if c in '.^$*+?()[]{}|/' else c for c in lit_token(rng)) + ')' + chr(92) + 's?=')})"
    return f"!{f}.starts_with({q(t)})"


def lit_rules(rng: random.Random, n: int):
    rules = []
    for k in range(n):
        e = lit_pred(rng)
        s = rng.randint(0, 5)
        if s == 0:
            e = e + " && " + lit_pred(rng)
        elif s == 1:
            e = e + " || " + lit_pred(rng)
        elif s == 2:
            e = "(" + e + " || " + lit_pred(rng) + ") && " + lit_pred(rng)
        rules.append((f"r{k}", e, fuzz_actions(rng)))
    return rules


def lit_field(rng: random.Random, lo=0, hi=6) -> str:
    parts = []
    for _ in range(rng.randint(lo, hi)):
        k = rng.random()
        if k < 0.45:
            parts.append(lit_token(rng))
        elif k < 0.55:
            t = rng.choice(TOKENS)
            parts.append(t.swapcase())
        elif k < 0.7:
            parts.append(str(rng.randint(0, 99999)))
        else:
            parts.append(rstr(rng, 1, 9, "abcdefxyz/.=-_ %"))
    return "".join(parts)


def lit_requests(rng: random.Random, n: int):
    reqs = []
    for _ in range(n):
        path = lit_field(rng, 0, 5)
        ua = lit_field(rng, 1, 4) if rng.random() < 0.95 else ""
        hdrs = {h: lit_field(rng, 0, 3) for h in ["x-a", "cookie", "referer", "x-unused"] if rng.random() < 0.6}
        reqs.append(Request(host=lit_field(rng, 0, 2), url=path + ("?" + lit_field(rng, 0, 4) if rng.random() < 0.6 else ""), path=path,
                            method=rng.choice(["GET", "POST"]), user_agent=ua[:255], ip=f"{rng.randint(1, 3)}.{rng.randint(0, 3)}.0.{rng.randint(0, 255)}",
                            remote_port=rng.randint(0, 65535), captcha_verified=rng.random() < 0.3, headers=hdrs or None))
    return reqs


REWRITTEN = {"rules": 0, "unsupported": 0}  # how often the fuzzers meet a rule the engine does not evaluate (the gap must be measured, not masked)


def as_the_engine_sees(rules, prog, allow=0):
    """The rule list with every rule the engine reports as unsupported (pwaf_program_rule_status; only possible with
    PWAF_OPT_LENIENT) replaced by one that never matches — the engine's documented behaviour for such a rule — so that the oracle can
    check everything else in the set. Since round 3 rules outside the column compiler's reach run in the residual interpreter, so
    the fuzz grammars produce NO such rule: `allow` is how many a test tolerates (default none)."""
    bad = set(prog.unsupported_rules(len(rules)))
    REWRITTEN["rules"] += len(rules)
    REWRITTEN["unsupported"] += len(bad)
    assert len(bad) <= allow, [(i, prog.rule_status(i)[1]) for i in sorted(bad)]
    return [(n, "false" if i in bad else e, a) for i, (n, e, a) in enumerate(rules)], bad
