#!/usr/bin/env python
"""One training step as a timeline from a rocprofv3 rocpd kernel trace: per kernel start offset, duration, queue.
usage: python tools/rocpd_timeline.py <results.db> [step_index=5] -- steps are delimited by k_adamw dispatches."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
step = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = cur.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if "k_adamw" in r[0]]
lo, hi = ends[step] + 1, ends[step + 1] + 1
t0 = rows[lo][1]
queues = {}
busy_until = {}
print(f"# step {step}: {hi - lo} kernels, {(rows[hi - 1][2] - t0) / 1e3:.1f} us from first start to last end; columns: start_us dur_us gap_us queue kernel")
for n, s, e, q in rows[lo:hi]:
    qi = queues.setdefault(q, len(queues))
    m = re.search(r"\b(k_[a-z_0-9]+)", n)
    short = m.group(1) if m else n[:40]
    gap = (s - busy_until.get(q, s)) / 1e3
    busy_until[q] = e
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.1f}  q{qi}  {'    ' * qi}{short}")
