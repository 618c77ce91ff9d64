#!/usr/bin/env python3
"""The oracle's Unicode regex semantics against CPython `re` in STR mode at scale (the generator of tests/test_oracle.py:
test_regex_unicode_matches_cpython_re_in_str_mode over many seeds; the alphabet keeps to code points where the two \\w / \\s definitions
agree). usage: python tools/regex_unicode_crosscheck.py <first seed> <last seed>   (500 pairs per seed; 0 mismatches expected)"""
import sys, random, re, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_oracle as T
from oracle import pyoracle
lo, hi = int(sys.argv[1]), int(sys.argv[2])
alphabet = ["a", "s", "S", "k", "K", "ſ", "K", "é", "É", "σ", "ς", "Σ", "€", "٣", " ", " ", " ", ".", "_", "1", "\n", "\U0001F600"]
bad = n = 0; t0 = time.time()
for seed in range(lo, hi):
    rng = random.Random(99_000_000 + seed)
    for _ in range(50):
        pat = T._rand_uregex(rng)
        try: pyre = re.compile(T._to_python(pat).decode())
        except re.error: continue
        for _ in range(10):
            hay = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 9)))
            if hay == "" and "\\B" in pat: continue
            want = pyre.search(hay) is not None
            try: got = pyoracle.regex_is_match(pat, hay.encode())
            except pyoracle.OracleError: break
            n += 1
            if got != want:
                bad += 1; print("MISMATCH", repr(pat), repr(hay), got, want, flush=True)
print("done", lo, hi, "pairs", n, "mismatches", bad, "time", round(time.time()-t0,1))
