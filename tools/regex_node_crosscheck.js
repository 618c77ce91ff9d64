#!/usr/bin/env node
// A THIRD implementation for the oracle's Unicode regex semantics (VERDICT r5 #5c): ECMAScript RegExp with the `u` flag (V8 / ICU — shares nothing
// with oracle/oracle_regex.cpp, CPython `re` or perl). stdin: a JSON array of [pattern, flags, haystack]; stdout: a JSON array of booleans
// (RegExp(pattern, flags).test(haystack)) or the string "error: .." for a pattern V8 rejects. The caller (tests/test_oracle.py) only sends
// constructs whose meaning coincides in both dialects: literals, `.` under the s flag, classes of explicit ranges and \p{..} / \P{..},
// negated classes, quantifiers, groups, alternation, ^ $ without the m flag, and the i flag on literals and positive ranges (simple case
// folding in both). NOT sent: \d \w \s \b (ASCII-only in ECMAScript's u mode), `.` without s (ECMAScript also excludes \r, U+2028, U+2029),
// case-insensitive negated / property classes (ECMAScript's pre-`v`-flag semantics differ).
'use strict';
let buf = '';
process.stdin.setEncoding('utf8');
process.stdin.on('data', (d) => { buf += d; });
process.stdin.on('end', () => {
  const out = JSON.parse(buf).map(([p, f, h]) => {
    try { return new RegExp(p, f).test(h); } catch (e) { return 'error: ' + e.message; }
  });
  process.stdout.write(JSON.stringify(out));
});
