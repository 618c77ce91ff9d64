#!/usr/bin/env python3
"""L2 -> fabric read requests per product-kernel launch by size (rocprofv3 --pmc TCC_EA0_RDREQ*_sum pass of tools/closing_r5.sh).

FETCH_SIZE is the request count x 64 B whatever the requests' sizes (/opt/skills/guides/MI355X_MICROARCH.md, HBM section: calibrated x2
for wide streaming reads = 128-B requests; other patterns are to be calibrated). This prints, per kernel, the requests of the last
pipeline pass by counter, so that a kernel of scattered 16-byte gathers (ipres_kernel) can be priced with its own request mix."""
import collections
import csv
import glob
import sys

per = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "pwaf::" in r["Kernel_Name"] or "rvm_jit" in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
names = sorted({c for d in per.values() for c in d})
print("kernel".ljust(44) + "".join(n.rjust(28) for n in names))
for k, d in per.items():
    print(k[:43].ljust(44) + "".join(f"{sorted(d[n])[-1][1]:28.4g}" if d[n] else " " * 28 for n in names))
