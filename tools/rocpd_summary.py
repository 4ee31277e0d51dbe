#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into per-kernel stats (CSV on stdout):
name, calls, total_ms, avg_us, min_us, max_us, percent.  Usage: rocpd_summary.py results.db"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    q = f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id"
    stats = defaultdict(list)
    for name, st, en in cur.execute(q):
        stats[name].append((en - st) / 1e3)  # ns -> us
    total = sum(sum(v) for v in stats.values())
    print("kernel,calls,total_ms,avg_us,min_us,max_us,percent")
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print(f"\"{name}\",{len(v)},{sum(v) / 1e3:.3f},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f},{100 * sum(v) / total:.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
