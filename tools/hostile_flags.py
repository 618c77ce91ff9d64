#!/usr/bin/env python3
"""How often the bigram prefilters of a synthetic config fire on its benign and on its HOSTILE stream (numpy model of filter_kernel, tables
tuned on a benign sample like bench.py does): candidate requests and completed windows (= the confirm tier's work) per request and pass.
Usage: python tools/hostile_flags.py [config] [n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import table_walker  # noqa: E402
from pingoo_amd.engine import CompiledProgram  # noqa: E402
from synth import pysynth  # noqa: E402


def hits(g, data, off):
    """windows completed at every arena position (bool array) under pass g's filter"""
    stride = int(g.get("f_stride", 1))
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
    t = (pos - start + (start % stride)) // stride
    sampled = (pos % stride == 0) & ((pos + 1) < endb)
    init = int(g["f_init"])
    top = (m >> 24) & 0xFF
    for j in range(1, 4):
        sh = j * stride
        shifted = np.concatenate([np.zeros(sh, dtype=np.uint64), m[:-sh]])
        top |= np.where(t >= j, (shifted >> (8 * (3 - j))) & 0xFF, 0).astype(np.uint64)
    for tt in range(0, 3):
        top |= np.where(t == tt, (init >> (8 * (3 - (tt + 1)))) & 0xFF, 0).astype(np.uint64)
    return ((top & 0xFF) != 0xFF) & sampled


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    wl = pysynth.Workload(cfg)
    prog = CompiledProgram(wl.rules, wl.lists, wl.geoip)
    prog.tune(wl.batch(10_000_000, 32768))
    t = table_walker.Tables(prog)
    names = ["host", "url", "path", "method", "user_agent"]
    tot = {}
    for label, adv in (("benign", False), ("hostile", True)):
        b = wl.batch(0, n, adversarial=adv)
        tot[label] = [0, 0]
        for gi, g in enumerate(t.groups):
            if "f_table" not in g:
                continue
            f = g["field"]
            cols = (b.data[f], b.offsets[f]) if f < 5 else b.headers.get(wl.header_names[f - 5])
            if cols is None:
                continue
            data, off = cols[0], cols[1].astype(np.int64)
            h = hits(g, np.concatenate([data, np.zeros(8, np.uint8)]), off)
            csum = np.concatenate([[0], np.cumsum(h)])
            o = np.minimum(off, len(h))
            per = csum[o[1:]] - csum[o[:-1]]
            chunks = len(np.unique(np.nonzero(h)[0] // 16))
            tot[label][0] += int((per > 0).sum())
            tot[label][1] += chunks
            if cfg != 5:
                print(f"{label:<8} pass {gi} {names[f] if f < 5 else 'hdr' + str(f - 5):<10} stride {g.get('f_stride')}: candidates {100 * (per > 0).mean():6.2f} % of requests, "
                      f"{h.sum() / n:.3f} windows and {chunks / n:.3f} flagged chunks per request = {100.0 * chunks / max(1, (int(off[-1]) + 15) // 16):.1f} % of the arena's chunks")
        print(f"{label}: {tot[label][0] / n:.3f} (request, pass) candidates and {tot[label][1] / n:.3f} flagged chunks per request over all filtered passes")


if __name__ == "__main__":
    main()
