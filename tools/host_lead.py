#!/usr/bin/env python
"""How far ahead of the GPU is the host?  rocprofv3 --kernel-trace --hip-runtime-trace db -> for every kernel of one step: the time its
launch call returned on the host, the time it started on the GPU, and the difference (lead).  A lead near zero = the GPU waited for
the host at that point.
usage: python tools/host_lead.py <results.db> [step_from_end=3]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
cols = lambda t: [r[1] for r in db.execute(f"pragma table_info({t})")]
kd, rg, ev, st_ = T("rocpd_kernel_dispatch"), T("rocpd_region"), T("rocpd_event"), T("rocpd_string")
ks = T("rocpd_info_kernel_symbol")
names = dict(db.execute(f"select id, string from {st_}"))
# API regions keyed by their event's correlation id
ecols = cols(ev)
corr = "stack_id" if "stack_id" in ecols else None   # (rocprofv3 7.x: the dispatch's event carries the stack id of the API call that enqueued it; correlation_id is 0)
if not corr:
    print("no correlation id column:", ecols)
    sys.exit(0)
api = {}
for s, e, nid, c in db.execute(f"select r.start, r.end, r.name_id, e.{corr} from {rg} r join {ev} e on r.event_id = e.id"):
    n = names.get(nid, "")
    if "Launch" in n or "hipModuleLaunch" in n or "ExtLaunch" in n:
        api[c] = (s, e, n)
kern = []
for s, e, kid, q, c in db.execute(f"select k.start, k.end, k.kernel_id, k.queue_id, e.{corr} from {kd} k join {ev} e on k.event_id = e.id order by k.start"):
    kern.append((s, e, kid, q, c))
kname = dict(db.execute(f"select id, kernel_name from {ks}")) if "kernel_name" in cols(ks) else {}
ends = [i for i, k in enumerate(kern) if "k_adamw" in kname.get(k[2], "")]
if len(ends) < back + 1:
    print("steps found:", len(ends)); sys.exit(0)
a, b = ends[-back - 1] + 1, ends[-back] + 1
t0 = kern[a][0]
print(f"step with {b - a} kernels; columns: gpu start us | lead us (gpu start - host launch return) | queue | kernel")
missing = 0
for s, e, kid, q, c in kern[a:b]:
    if c not in api:
        missing += 1
        continue
    hs, he, n = api[c]
    print(f"{(s - t0) / 1e3:9.1f} {(s - he) / 1e3:9.1f}  q{q}  {kname.get(kid, '?')[:60]}")
print("kernels without a matched launch call:", missing)
