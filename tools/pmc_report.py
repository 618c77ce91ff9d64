#!/usr/bin/env python3
"""Merges the per-pass rocprofv3 counter CSVs written by tools/pmc_scan.sh into one table per product-kernel dispatch."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
data = collections.defaultdict(dict)
for d in sorted(glob.glob(f"{root}/pmc_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        if "pwaf" in r["Kernel_Name"] or "rvm_jit" in r["Kernel_Name"]:  # (rvm_jit_kernel: the specialized residual program, extern "C")
            data[(int(r["Dispatch_Id"]), r["Kernel_Name"].replace("void ", "")[:28])][r["Counter_Name"]] = float(r["Counter_Value"])
rows = sorted(data.items())[-26:]
cols = sorted({c for _, v in rows for c in v})
print("dispatch kernel " + " ".join(c.replace("SQ_", "") for c in cols))
for (did, k), v in rows:
    print(did, k, " ".join("%.3g" % v.get(c, float("nan")) for c in cols))
