import sqlite3, sys, glob
from collections import defaultdict
pat = sys.argv[2] if len(sys.argv) > 2 else 'fused'
for path in sorted(glob.glob(sys.argv[1])):
    con = sqlite3.connect(path); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    pm = [t for t in tabs if t.startswith("rocpd_pmc_event")]
    if not pm: continue
    pm = pm[0]; info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    acc = defaultdict(lambda: defaultdict(float)); dur = {}
    for k, c, v, did, st, en in cur.execute(f"select s.display_name, i.name, p.value, d.id, d.start, d.end from {pm} p join {info} i on p.pmc_id = i.id join {disp} d on d.event_id = p.event_id join {sym} s on d.kernel_id = s.id"):
        if pat in k: acc[c][did] += v; dur[did] = (en - st) / 1e3
    for c, d in acc.items():
        vals = list(d.values()); a = sum(vals) / len(vals)
        print(f"{c:28s} {a:.4g}  (avg dur {sum(dur.values())/len(dur):.0f} us)")
