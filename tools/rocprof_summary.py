#!/usr/bin/env python3
"""Turns a rocprofv3 `--kernel-trace --stats` result database into the text summary committed under profiles/.
usage: tools/rocprof_summary.py <results.db> [title]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    c = sqlite3.connect(db)
    print(f"# {title}")
    print("# source: rocprofv3 --kernel-trace --stats (views top_kernels / kernels of the rocpd database)")
    print(f"{'kernel':<60} {'calls':>6} {'total_ms':>12} {'avg_us':>12} {'pct':>7}")
    for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print(f"{name[:60]:<60} {calls:>6} {total / 1e3:>12.3f} {avg:>12.3f} {pct:>7.2f}")
    print()
    print("# per-launch detail of the product kernels (last pipeline pass in the trace)")
    rows = list(c.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count from kernels where name like '%pwaf::%' or name like '%rvm_jit%' order by start"))
    last_verdict = max(i for i, r in enumerate(rows) if "verdict" in r[0])
    prev = max([i for i, r in enumerate(rows[:last_verdict]) if "verdict" in r[0]] + [-1])
    print(f"{'kernel':<40} {'dur_us':>10} {'grid':>9} {'wg':>5} {'lds_B':>8} {'vgpr':>5} {'sgpr':>5}")
    for name, dur, gx, wx, lds, vg, sg in rows[prev + 1:last_verdict + 1]:
        print(f"{name[:40]:<40} {dur / 1e3:>10.1f} {gx:>9} {wx:>5} {lds:>8} {vg:>5} {sg:>5}")


if __name__ == "__main__":
    main()
