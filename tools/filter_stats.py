#!/usr/bin/env python3
"""Host-side look at the bigram prefilters of a synthetic config: which passes are filtered, their heads, and the candidate
rate on a sample of the synthetic stream (numpy model of filter_kernel). Usage: python tools/filter_stats.py [config] [n] [--tune]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pingoo_amd.engine import CompiledProgram
from synth import pysynth
import table_walker



def candidates(g, data, off):
    STRIDE = int(g.get("f_stride", 1))
    """numpy model of filter_kernel over one field arena (bigrams sampled every STRIDE bytes from each field's start): bool per request"""
    d = data[: off[-1] + 1].astype(np.uint32)
    d = d & ~((d >> 1) & 0x20)  # program.h: filter_fold
    p = d[:-1] | (d[1:] << 8)
    bins = ((p * int(g["f_mul"])) & 0xFFFF) >> 4
    m = g["f_table"][bins].astype(np.uint64)
    L = len(m)
    pos = np.arange(L)
    lens = np.diff(off)
    start = np.repeat(off[:-1], lens)[:L]
    endb = np.repeat(off[1:], lens)[:L]
    if len(start) < L:
        start = np.concatenate([start, np.full(L - len(start), off[-1])])
        endb = np.concatenate([endb, np.full(L - len(endb), off[-1])])
    t = (pos - start + (start % STRIDE)) // STRIDE  # sampled step index inside the field (grid: even arena offsets at stride 2)
    sampled = (pos % STRIDE == 0) & ((pos + 1) < endb)
    init = int(g["f_init"])
    top = (m >> 24) & 0xFF
    for j in range(1, 4):
        sh = j * STRIDE
        shifted = np.concatenate([np.zeros(sh, dtype=np.uint64), m[:-sh]])
        contrib = (shifted >> (8 * (3 - j))) & 0xFF
        top |= np.where(t >= j, contrib, 0).astype(np.uint64)
    for tt in range(0, 3):  # the field's first steps still see the initial state
        ib = (init >> (8 * (3 - (tt + 1)))) & 0xFF
        top |= np.where(t == tt, ib, 0).astype(np.uint64)
    hit = ((top & 0xFF) != 0xFF) & sampled
    csum = np.concatenate([[0], np.cumsum(hit)])
    o = np.minimum(off, L)
    return (csum[o[1:]] - csum[o[:-1]]) > 0


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    wl = pysynth.Workload(cfg)
    prog = CompiledProgram(wl.rules, wl.lists, wl.geoip, flags=16 if '--stride2' in sys.argv else 0)
    print(prog.stats())
    t = table_walker.Tables(prog)
    b = wl.batch(0, n)
    names = ["host", "url", "path", "method", "user_agent"]
    for gi, g in enumerate(t.groups):
        line = f"group {gi} field {names[g['field']] if g['field'] < 5 else 'hdr' + str(g['field'] - 5):<10} stride {g.get('f_stride', '-')} states {g['n_states']:>6} classes {g['n_classes']:>3} atoms {g['n_local']:>4}"
        if "f_table" in g:
            data, off = b.data[g["field"]], b.offsets[g["field"]].astype(np.int64)
            c = candidates(g, np.concatenate([data, np.zeros(8, np.uint8)]), off)
            zeros = [(int(g["f_table"][k]) ^ 0xFFFFFFFF) for k in range(0, 4096)]
            filled = sum(1 for z in zeros if z)
            line += f"  FILTER init {g['f_init']:08x} heads {[(h[0], h[1]) for h in g['f_heads']]} bins touched {filled}  candidates {c.mean() * 100:.2f} %"
        print(line)

main()

def crosscheck():
    wl = pysynth.Workload(3)
    prog = CompiledProgram(wl.rules, wl.lists, wl.geoip)
    t = table_walker.Tables(prog)
    b = wl.batch(0, 3000)
    for g in t.groups:
        if "f_table" not in g:
            continue
        data, off = b.data[g["field"]], b.offsets[g["field"]].astype(np.int64)
        c = candidates(g, np.concatenate([data, np.zeros(8, np.uint8)]), off)
        ref = np.array([t.filter_candidate(g, b.field_bytes(g["field"], i)) for i in range(3000)])
        # and soundness: any request whose DFA reports a non-head atom must be a candidate
        missed = 0
        heads = {h[2] for h in g["f_heads"]}
        for i in range(3000):
            cols = set()
            t.scan_field(g, b.field_bytes(g["field"], i), cols)
            if any((cc - g["atom_base"]) not in heads for cc in cols) and not ref[i]:
                missed += 1
        print("field", g["field"], "numpy==python:", bool((c == ref).all()), "cand", int(ref.sum()), "missed", missed)

if os.environ.get("CROSSCHECK"):
    crosscheck()
