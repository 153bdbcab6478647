#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counters (rocpd sqlite output) per kernel name.  usage: rocpd_pmc.py results.db [out.txt]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type='view'")]
    src = "counters_collection" if "counters_collection" in views else "pmc_events"
    cols = [r[1] for r in cur.execute(f"pragma table_info({src})")]
    ncol = "kernel_name" if "kernel_name" in cols else "name"
    ccol = "counter_name" if "counter_name" in cols else "pmc_name"
    vcol = "value" if "value" in cols else "counter_value"
    dcol = "dispatch_id" if "dispatch_id" in cols else "id"
    rows = cur.execute(f"select {ncol}, {ccol}, {dcol}, sum({vcol}) from {src} group by {ncol}, {ccol}, {dcol}").fetchall()
    agg = {}
    for name, ctr, _d, val in rows:
        short = re.sub(r"\(.*$", "", name)
        a = agg.setdefault((short, ctr), [0, 0.0])
        a[0] += 1
        a[1] += val
    lines = [f"{'kernel':64s} {'counter':28s} {'dispatches':>10s} {'sum':>18s} {'per_dispatch':>16s}"]
    for (k, c), (n, v) in sorted(agg.items(), key=lambda kv: (-kv[1][1], kv[0])):
        lines.append(f"{k[:64]:64s} {c:28s} {n:10d} {v:18.1f} {v / n:16.1f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
