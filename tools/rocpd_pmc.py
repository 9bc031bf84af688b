#!/usr/bin/env python
"""Average PMC counter value per kernel from a rocprofv3 rocpd sqlite database (--pmc run)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
print(cols)
rows = cur.execute("select kernel_name, counter_name, value from counters_collection").fetchall() if "kernel_name" in cols else []
agg = {}
for n, c, v in rows:
    if pat and pat not in n: continue
    short = re.sub(r"\(anonymous namespace\)::", "", n)[:80]
    a = agg.setdefault((short, c), [0, 0.0]); a[0] += 1; a[1] += v
for (n, c), (k, t) in sorted(agg.items()):
    print(f"{n:82s} {c:12s} calls={k:4d} avg={t/k:14.1f}")
