#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database: rocpd_pmc.py results.db"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    pm = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    dcols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    ev_col = "event_id" if "event_id" in dcols else "id"
    q = (f"select s.{name_col}, i.name, p.value, d.id from {pm} p join {info} i on p.pmc_id = i.id "
         f"join {disp} d on d.{ev_col} = p.event_id join {sym} s on d.kernel_id = s.id")
    acc = defaultdict(lambda: defaultdict(float))  # (kernel, counter) -> dispatch id -> sum over XCD/SE instances
    for kname, cname, val, did in cur.execute(q):
        acc[(kname, cname)][did] += val
    print("kernel,counter,dispatches,avg_value,units_note")
    for (kname, cname), per in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
        v = list(per.values())
        note = "KiB per dispatch" if cname in ("FETCH_SIZE", "WRITE_SIZE") else "per dispatch (summed over XCD/SE instances)"
        print(f"\"{kname}\",{cname},{len(v)},{sum(v) / len(v):.1f},{note}")


if __name__ == "__main__":
    main(sys.argv[1])
