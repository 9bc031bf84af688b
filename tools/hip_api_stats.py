#!/usr/bin/env python
"""HIP API host time of a rocprofv3 --hip-runtime-trace (rocpd sqlite): calls and microseconds per API per step.
usage: python tools/hip_api_stats.py <results.db> <steps>"""
import sqlite3
import sys

db, steps = sys.argv[1], int(sys.argv[2])
cur = sqlite3.connect(db).cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
reg = [n for n in names if n == "regions"] or [n for n in names if n.startswith("regions")]
if not reg:
    print("no regions view; tables:", names)
    sys.exit(0)
cols = [r[1] for r in cur.execute(f"pragma table_info({reg[0]})")]
rows = list(cur.execute(f"select name, count(*), sum(end - start) from {reg[0]} group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"HIP API host time per step ({steps} steps): {tot / 1e3 / steps:.1f} us over {sum(r[1] for r in rows) / steps:.0f} calls")
for n, c, t in rows[:25]:
    print(f"  {n:44s} {c / steps:8.1f} calls/step {t / 1e3 / steps:9.1f} us/step {t / 1e3 / c:7.2f} us/call")
