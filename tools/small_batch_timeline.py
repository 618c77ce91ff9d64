#!/usr/bin/env python3
"""Where the time of a SMALL host batch goes (the micro-batcher's unit of work): wall time per pwaf_evaluate_batch call of N requests of
config 3, and — under `rocprofv3 --kernel-trace --memory-copy-trace` — the device timeline of the last call. usage: small_batch_timeline.py [N] [calls]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pingoo_amd.engine import RuleEngine  # noqa: E402
from synth import pysynth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w = pysynth.Workload(3)
eng = RuleEngine(w.rules, w.lists, w.geoip)
eng.tune(w.batch(10_000_000, 32768))
b = w.batch(0, n)
for _ in range(20):
    eng.evaluate_batch(b)
lat = []
for _ in range(calls):
    t0 = time.perf_counter()
    eng.evaluate_batch(b)
    lat.append(time.perf_counter() - t0)
lat = np.array(lat) * 1e6
print(f"n={n}: pwaf_evaluate_batch wall time per call: p50 {np.percentile(lat, 50):.1f} us, p10 {np.percentile(lat, 10):.1f}, p90 {np.percentile(lat, 90):.1f} ({calls} calls, Python caller)")
eng.close()
