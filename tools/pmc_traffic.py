#!/usr/bin/env python3
"""HBM traffic per product-kernel launch from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh.

rocprofv3 reports both in KiB. Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts the
128-byte requests of a wide COALESCED read (16 B per lane, consecutive lanes) at 64 bytes, so it is doubled — "other access widths
are uncalibrated: calibrate on a known byte count in your own access pattern". filter_kernel's pattern is not that one: a lane reads
its own 64-byte segment (four 16-B loads, lanes 64 B apart), i.e. 64-byte requests, and its known byte count is the arena it streams
exactly once: the RAW counter equals the arena bytes (2.89 GB counted for 2.83 GB of arenas + offsets), so its factor is 1. Every
other kernel keeps the x2 of the guide (an upper bound for gathers). WRITE_SIZE is taken as reported (uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(f"{root}/pmc_{cname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "pwaf::" in r["Kernel_Name"] and r["Counter_Name"] == cname:
                per[r["Kernel_Name"].split("(")[0].replace("void ", "")][cname].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
out = {"commit": os.environ.get("PWAF_COMMIT", "?"), "unit": "bytes per launch, last pipeline pass of the trace",
       "fetch_correction": "x2 (gfx950: 128-B requests tallied at 64 B) except filter_kernel: x1, calibrated on the arena bytes it streams once (64-B requests: one segment per lane)", "kernels": {}}
passes = max(1, max((len(d["FETCH_SIZE"]) for k, d in per.items() if "verdict" in k), default=1))  # pipeline passes in the trace
for k, d in per.items():
    f = [v for _, v in sorted(d["FETCH_SIZE"])]
    w = [v for _, v in sorted(d["WRITE_SIZE"])]
    n = max(1, len(f) // passes) if len(f) >= passes else len(f)
    f, w = f[-n:], w[-n:]
    factor = 1 if "filter_kernel" in k else 2
    out["kernels"][k] = {"launches": len(f), "fetch_factor": factor, "fetch_bytes": [int(x * 1024 * factor) for x in f], "write_bytes": [int(x * 1024) for x in w]}
print(json.dumps(out, indent=1))
