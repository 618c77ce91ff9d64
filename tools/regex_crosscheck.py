#!/usr/bin/env python3
"""Cross-checks the oracle's regex engine (oracle/oracle_regex.cpp: the CPU restatement of `regex 1.12.2` is_match, Cargo.lock:1694-1700)
against CPython `re` on the FULL regex rule sets of the synthetic configs — every `matches("...")` pattern of BASELINE configs[2] and
configs[4] — over the field values of benign AND adversarial requests of the same generator (VERDICT r2 #8a: the fuzz grammar of
tests/test_oracle.py is not the rule set the headline number is measured on).

    python tools/regex_crosscheck.py [--configs 3 5] [--requests 3000]

The patterns use only syntax on which the regex crate and CPython agree once `$` is written as `\\Z` (no multi-line flags appear in
the rule sets): (?i), classes, escapes, counted repetitions, alternation, \\b, \\s. Haystacks are bytes. Exit status 1 on any mismatch."""
from __future__ import annotations

import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FIELD_IDS = {"host": 0, "url": 1, "path": 2, "method": 3, "user_agent": 4}
CALL = re.compile(r'http_request\.(?:(host|url|path|method|user_agent)|headers\["([^"]+)"\])\.matches\("((?:[^"\\]|\\.)*)"\)')


def unquote(s: str) -> str:
    """CEL string literal body -> the pattern text (the generator only escapes backslash and double quote)."""
    out, i = "", 0
    while i < len(s):
        if s[i] == "\\" and i + 1 < len(s):
            out += s[i + 1]
            i += 2
        else:
            out += s[i]
            i += 1
    return out


def to_python(pat: str) -> bytes:
    """`$` outside classes / escapes -> \\Z (Rust's `$` without (?m) matches only at the very end; CPython's also before a final newline)."""
    out, i, in_class = "", 0, False
    while i < len(pat):
        ch = pat[i]
        if ch == "\\":
            out += pat[i:i + 2]
            i += 2
            continue
        if in_class:
            if ch == "]":
                in_class = False
        elif ch == "[":
            in_class = True
        elif ch == "$":
            out += "\\Z"
            i += 1
            continue
        out += ch
        i += 1
    return out.encode()


def crosscheck(config: int, n_requests: int, verbose: bool = True):
    from oracle import pyoracle
    from synth import pysynth

    w = pysynth.Workload(config)
    pats = []
    for name, expr, _ in w.rules:
        if expr is None:
            continue
        for m in CALL.finditer(expr):
            pats.append((m.group(1) or "", m.group(2) or "", unquote(m.group(3))))
    pats = sorted(set(pats))
    assert pats, "no regex rules in this config"
    batches = [w.batch(0, n_requests), w.batch(0, n_requests, adversarial=True)]
    checked = mismatches = 0
    for field, header, pat in pats:
        assert "(?m" not in pat and "(?s" not in pat, pat
        pyre = re.compile(to_python(pat))
        for b in batches:
            hays = set()
            for i in range(b.n):
                hays.add(b.field_bytes(FIELD_IDS[field], i) if field else b.header_bytes(header, i))
            for hay in hays:
                want = pyre.search(hay) is not None
                got = pyoracle.regex_is_match(pat, hay)
                checked += 1
                if got != want:
                    mismatches += 1
                    if verbose:
                        print(f"MISMATCH config {config} pattern {pat!r} haystack {hay!r}: oracle {got}, re {want}")
    if verbose:
        print(f"config {config}: {len(pats)} distinct regex patterns x field values of {n_requests} benign + {n_requests} adversarial requests: {checked} (pattern, haystack) pairs, {mismatches} mismatches")
    return len(pats), checked, mismatches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, nargs="+", default=[3, 5])
    ap.add_argument("--requests", type=int, default=3000)
    a = ap.parse_args()
    bad = 0
    for c in a.configs:
        bad += crosscheck(c, a.requests)[2]
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
