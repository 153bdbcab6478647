#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as a per-kernel stats table
(what `--stats` prints as kernel_stats.csv with the csv output format).  usage: rocpd_stats.py results.db [out.txt]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        short = re.sub(r"\(.*$", "", name)
        a = agg.setdefault(short, [0, 0, 10**18, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = [f"{'kernel':70s} {'calls':>7s} {'total_ms':>11s} {'avg_us':>11s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k[:70]:70s} {a[0]:7d} {a[1] / 1e6:11.3f} {a[1] / a[0] / 1e3:11.2f} {a[2] / 1e3:10.2f} {a[3] / 1e3:10.2f} {100 * a[1] / tot:6.2f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
