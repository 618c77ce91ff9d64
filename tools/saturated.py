#!/usr/bin/env python3
"""The SATURATED stream: what an attacker who has read the rule set — and the engine's tables: they are a function of it —
sends to make the confirm tier work on every byte. url, path and User-Agent are filled to their caps with tokens that COMPLETE A WINDOW of
the pass's bigram filter without being the literal they imitate (the literal minus its first or last byte ...), chosen with the numpy
model of filter_kernel over the engine's own (tuned) filter tables: a token is kept only if the model says it flags. Host, method and
the client columns are the benign stream's.

    batch, info = saturated_batch(wl, program, n)      # program: CompiledProgram / engine.program (tuned or not)

`info` says how many 16-byte chunks of each arena the model flags (the bench leg reports it). Input generation only: nothing here is on
the product path. Usage as a script: python tools/saturated.py [config] [n]  -> prints the flagged fractions."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

CAPS = {1: 500, 2: 120, 4: 250}  # url / path / user_agent bytes (the reference caps host and User-Agent at 256: http_utils.rs:20-21)
SEP = {1: b"&", 2: b"/", 4: b" "}
FIELD_NAMES = {"url": 1, "path": 2, "user_agent": 4}
LIT = re.compile(r'http_request\.(url|path|user_agent)(?:\.(?:contains|starts_with|ends_with)\(|\s*==\s*)"((?:[^"\\]|\\.)*)"')


def _hits(g, data, off):
    from hostile_flags import hits  # (the numpy model of filter_kernel)

    return hits(g, data, off)


def _tokens(wl, tables, field):
    """near misses of the rule literals on `field` that complete a window of the field's filtered pass"""
    lits = set()
    for _, expr, _ in wl.rules:
        for f, lit in LIT.findall(expr or ""):
            if FIELD_NAMES[f] == field:
                lits.add(lit.encode().decode("unicode_escape").encode("latin1"))
    cands = set()
    for lit in lits:
        if len(lit) >= 5:
            cands.update((lit[:-1], lit[1:-1]))  # (not lit[1:]: behind the separator it may spell the literal again)
    cands = sorted(c for c in cands if len(c) >= 4 and not any(l in c for l in lits))
    passes = [g for g in tables.groups if g["field"] == field and "f_table" in g]
    if not cands or not passes:
        return [], sorted(lits)
    g = passes[0]
    sep = SEP[field]
    blob = sep.join(cands) + sep
    lens = np.array([len(c) + 1 for c in cands], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    h = _hits(g, np.frombuffer(blob + b"\0" * 16, dtype=np.uint8), off)
    csum = np.concatenate([[0], np.cumsum(h)])
    o = np.minimum(off, len(h))
    keep = (csum[o[1:]] - csum[o[:-1]]) > 0
    return [c for c, k in zip(cands, keep) if k], sorted(lits)


def saturated_batch(wl, program, n, block=1 << 19, seed=0x5A7):
    import table_walker
    from pingoo_amd import RequestBatch, _abi

    tables = table_walker.Tables(program)
    rng = np.random.default_rng(seed)
    block = min(block, n)
    base = wl.batch(0, block)
    data, offs = list(base.data), list(base.offsets)
    info = {}
    for field in (1, 2, 4):
        toks, lits = _tokens(wl, tables, field) or ([], [])
        if not toks:
            continue
        sep, cap = SEP[field], CAPS[field]
        pool = []
        for _ in range(4096):  # distinct field values; a request draws one per field
            s = b"/" if field != 4 else b""
            while len(s) < cap:
                s += toks[int(rng.integers(len(toks)))] + sep
            s = s[:cap].rstrip(b"/") if field == 2 else s[:cap]
            for _ in range(8):  # near misses only: a value in which two tokens happen to spell a rule literal is redrawn piece by piece
                bad = [l for l in lits if l in s]
                if not bad:
                    break
                for l in bad:
                    s = s.replace(l, l[:-1] + sep)[:cap]
            pool.append(s)
        pick = rng.integers(len(pool), size=block)
        lens = np.array([len(p) for p in pool], dtype=np.int64)[pick]
        o = np.zeros(block + 1, dtype=np.uint32)
        o[1:] = np.cumsum(lens).astype(np.uint32)
        arena = np.frombuffer(b"".join(pool[int(k)] for k in pick) + b"\0" * _abi.ARENA_PAD, dtype=np.uint8).copy()
        data[field], offs[field] = arena, o
        g = [x for x in tables.groups if x["field"] == field and "f_table" in x][0]
        m = min(block, 4096)
        h = _hits(g, arena[: int(o[m]) + 16], o[: m + 1].astype(np.int64))
        info[["", "url", "path", "", "user_agent"][field]] = {"tokens": len(toks), "flagged_chunk_fraction_model": round(len(np.unique(np.nonzero(h)[0] // 16)) / max(1, (int(o[m]) + 15) // 16), 3)}
    b = RequestBatch(data, offs, base.ip, base.ip_is_v6, base.port, base.flags, base.asn, base.country, base.headers)
    times = max(1, n // block)
    return (b.tile(times) if times > 1 else b), info


if __name__ == "__main__":
    from pingoo_amd.engine import CompiledProgram
    from synth import pysynth

    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    wl = pysynth.Workload(cfg)
    prog = CompiledProgram(wl.rules, wl.lists, wl.geoip)
    prog.tune(wl.batch(10_000_000, 32768))
    batch, info = saturated_batch(wl, prog, n, block=n)
    print(info, batch.n, [len(batch.field_bytes(f, 0)) for f in (1, 2, 4)], batch.field_bytes(4, 0)[:80])
