#!/usr/bin/env python
"""Per (kernel, grid) average dispatch duration from a rocprofv3 rocpd sqlite trace."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, (end-start) from kernels").fetchall() if "grid_x" in cols else None
if rows is None:
    print(cols); sys.exit(0)
agg = {}
for n, gx, gy, gz, wx, d in rows:
    if pat and pat not in n: continue
    short = re.sub(r"\(anonymous namespace\)::", "", n)[:70]
    k = (short, gx // max(wx,1), gy, gz)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
for k, (c, t) in sorted(agg.items(), key=lambda kv: kv[0]):
    print(f"{k[0]:72s} grid=({k[1]},{k[2]},{k[3]}) calls={c:4d} avg_us={t/c/1e3:9.2f}")
