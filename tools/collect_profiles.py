#!/usr/bin/env python3
"""Copies what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/<name>_{kernel_stats.txt,traffic.json,counters.txt} —
after checking that the profile was taken at the commit that last touched the kernels (profiles must describe HEAD).
usage: python tools/collect_profiles.py <tag> <name>      e.g.  r5_c3 r5_c3_10M"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
head = subprocess.run(["git", "log", "-1", "--format=%h", "--", "pingoo_amd/csrc"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
traffic = json.load(open(os.path.join(src, "traffic.json")))
if traffic.get("commit") != head:
    sys.exit(f"profile {tag} was taken at commit {traffic.get('commit')!r}, the kernels' last commit is {head!r}: re-take it")
for f in ("kernel_stats.txt", "traffic.json", "counters.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(ROOT, "profiles", f"{name}_{f}"))
        print("profiles/" + f"{name}_{f}")
