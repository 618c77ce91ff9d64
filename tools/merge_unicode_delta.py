#!/usr/bin/env python3
"""oracle/unicode_data.inc = perl's Unicode 13.0 tables (tools/gen_unicode_tables.pl) + what node's ICU (Unicode 14.0,
tools/gen_unicode_tables_node.js) says about the code points that 13.0 leaves UNASSIGNED.

Why: the device compiler reads node's tables (pingoo_amd/csrc/unicode_data.inc), the oracle perl's — two independent implementations of
the Unicode Character Database, so that a wrong script range / White_Space member / case-folding orbit on one side shows up in every
GPU-vs-oracle and table-vs-oracle test (the two files used to be the same file). On the code points both versions
assign (all of 13.0: 99.4 % of 14.0's assigned code points) the oracle keeps perl's answer; on the 838 code points 14.0 added it has no
answer of its own and takes node's, so that both sides describe the same Unicode version (D19: 14.0 vs the crate's newer one).

    python tools/merge_unicode_delta.py PERL13.inc NODE14.inc > oracle/unicode_data.inc      # merge
    python tools/merge_unicode_delta.py --compare A.inc B.inc [--outside-cn-of PERL13.inc]   # differences, per table, as code point counts
"""
import re
import sys


def parse(path):
    text = open(path).read()
    rng = [(int(a, 16), int(b, 16)) for a, b in re.findall(r"\{0x([0-9A-F]+),0x([0-9A-F]+)\}", text[text.index("kUniRanges"):text.index("struct UniTable")])]
    tables = []
    for kind, names, first, count in re.findall(r'\{(\d), "([^"]*)", (\d+), (\d+)\},', text[text.index("kUniTables"):text.index("kUniFold")]):
        tables.append((int(kind), names.split("|"), rng[int(first):int(first) + int(count)]))
    fold = [(int(a, 16), int(b, 16)) for a, b in re.findall(r"\{0x([0-9A-F]+),0x([0-9A-F]+)\}", text[text.index("kUniFold"):])]
    return tables, fold


def to_set(ranges):
    s = set()
    for lo, hi in ranges:
        s.update(range(lo, hi + 1))
    return s


def to_ranges(s):
    out = []
    for cp in sorted(s):
        if out and out[-1][1] == cp - 1 and cp != 0xE000:
            out[-1][1] = cp
        else:
            out.append([cp, cp])
    return out


def key(t):
    return (t[0], min(t[1], key=lambda n: (-len(n), n)))  # (kind, the longest alias: the two generators list a table's aliases in different orders)


def unassigned(tables):
    for kind, names, r in tables:
        if kind == 0 and "cn" in names:
            return to_set(r)
    raise SystemExit("no Cn table")


def emit(header, tables, fold):
    out = [header, "// kind: 0 = General_Category value, 1 = Script value, 2 = binary property. Names are matched loosely (case, '_', '-', ' ' ignored)."]
    flat, index = [], []
    for kind, names, r in tables:
        index.append('{%d, "%s", %d, %d},' % (kind, "|".join(names), len(flat), len(r)))
        flat += r
    out.append("static const unsigned kUniRanges[][2] = {")
    for i in range(0, len(flat), 8):
        out.append("".join("{0x%X,0x%X}," % tuple(x) for x in flat[i:i + 8]))
    out.append("};")
    out.append("struct UniTable { int kind; const char *names; unsigned first, count; };")
    out.append("static const UniTable kUniTables[] = {")
    out += index
    out.append("};")
    out.append("// (cp, other member of its simple case folding orbit), sorted by cp")
    out.append("static const unsigned kUniFold[][2] = {")
    for i in range(0, len(fold), 8):
        out.append("".join("{0x%X,0x%X}," % x for x in fold[i:i + 8]))
    out.append("};")
    return "\n".join(out) + "\n"


# What CHANGED between 13.0 and 14.0 on code points 13.0 already assigns (the comparison of the two generators' outputs finds exactly these,
# tests/test_oracle.py: test_unicode_tables_come_from_two_independent_sources): U+1734 HANUNOO SIGN PAMUDPOD Mn -> Mc, U+16FE2 / U+16FE3
# (Old Chinese marks) Common -> Han. The oracle takes node's answer for them: both sides describe 14.0.
CHANGED_IN_14 = {0x1734, 0x16FE2, 0x16FE3}
# ... and three case-folding orbits that V8's closure has and perl's 13.0 simple folding has not: U+0390 / U+1FD3, U+03B0 / U+1FE3, U+FB05 / U+FB06.
# CaseFolding.txt gives them simple foldings since Unicode 15.1, i.e. regex-syntax 0.8.8's tables (the reference's, Cargo.lock:1717-1720) fold them.
FOLDS_SINCE_15_1 = {(0x390, 0x1FD3), (0x1FD3, 0x390), (0x3B0, 0x1FE3), (0x1FE3, 0x3B0), (0xFB05, 0xFB06), (0xFB06, 0xFB05)}


def main():
    if sys.argv[1] == "--compare":
        a, fa = parse(sys.argv[2])
        b, fb = parse(sys.argv[3])
        mask = unassigned(parse(sys.argv[5])[0]) if len(sys.argv) > 5 and sys.argv[4] == "--outside-cn-of" else set()
        da, db = {key(t): t for t in a}, {key(t): t for t in b}
        n_diff = 0
        for k in sorted(set(da) | set(db)):
            sa, sb = (to_set(da[k][2]) if k in da else set()), (to_set(db[k][2]) if k in db else set())
            d = (sa ^ sb) - mask
            if d or (k in da) != (k in db) and (sa | sb) - mask:
                n_diff += len(d)
                print("table", k, "differs on", len(d), "code points, e.g.", ["%X" % x for x in sorted(d)[:6]])
        d = {p for p in set(fa) ^ set(fb) if p[0] not in mask and p[1] not in mask}
        n_diff += len(d)
        if d:
            print("case folding differs on", len(d), "pairs, e.g.", sorted(d)[:6])
        print("differences:", n_diff)
        return 1 if n_diff else 0
    perl, pfold = parse(sys.argv[1])
    node, nfold = parse(sys.argv[2])
    cn13 = unassigned(perl) | CHANGED_IN_14  # (where node's answer is taken)
    dn = {key(t): t for t in node}
    merged = []
    seen = set()
    for kind, names, r in perl:
        k = key((kind, names, r))
        seen.add(k)
        s = to_set(r) - cn13
        if k in dn:
            s |= to_set(dn[k][2]) & cn13
        merged.append((kind, names, to_ranges(s)))
    for kind, names, r in node:  # tables 13.0 does not have (the scripts 14.0 added): what node says, which lies in 13.0's unassigned space
        if key((kind, names, r)) not in seen:
            merged.append((kind, names, to_ranges(to_set(r) & cn13)))
    fold = sorted(set(pfold) | {p for p in nfold if p[0] in cn13 or p[1] in cn13 or p in FOLDS_SINCE_15_1})
    sys.stdout.write(emit("// GENERATED by tools/merge_unicode_delta.py: perl's Unicode Character Database (tools/gen_unicode_tables.pl, Unicode 13.0.0) on every code point 13.0 assigns; "
                          "node's ICU (tools/gen_unicode_tables_node.js, Unicode 14.0) on the code points 13.0 leaves unassigned. Data, not code.", merged, fold))
    return 0


if __name__ == "__main__":
    sys.exit(main())
