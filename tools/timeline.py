#!/usr/bin/env python
"""Per-step timeline of a rocprofv3 kernel trace (rocpd sqlite): which queue runs what, when, and how full the chip is.

usage: python tools/timeline.py <results.db> <out.txt> [step index from the end, default 3]
Steps are delimited by the k_adamw launch that ends each one.  For the chosen step the file lists every kernel
(start offset, duration, queue, workgroups, name) and, for the middle steps, the span, the union of busy time, the time
per queue and the time during which fewer than 256 workgroups' worth of kernels were resident (chip under-filled)."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"\b(k_[a-z_0-9]+)(<[^(]*)?", n)
    return (m.group(1) + (m.group(2) or "")[:40]) if m else n[:50]


def main():
    db, out = sys.argv[1], sys.argv[2]
    pick = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, queue_id, start, end, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z) "
                            "from kernels order by start"))
    ends = [i for i, r in enumerate(rows) if "k_adamw" in r[0]]
    steps = [rows[ends[i] + 1:ends[i + 1] + 1] for i in range(len(ends) - 1)]
    with open(out, "w") as f:
        spans, unions, under = [], [], []
        perq = {}
        for st in steps[len(steps) // 2:]:
            t0, t1 = st[0][2], max(r[3] for r in st)
            spans.append(t1 - t0)
            ev = sorted([(r[2], 1, r[4]) for r in st] + [(r[3], -1, r[4]) for r in st])
            busy = wg = 0
            u = lowfill = 0
            last = t0
            for t, d, g in ev:
                if busy > 0:
                    u += t - last
                    if wg < 256:
                        lowfill += t - last
                last = t
                busy += d
                wg += d * g
            unions.append(u)
            under.append(lowfill)
            for r in st:
                perq[r[1]] = perq.get(r[1], 0) + (r[3] - r[2])
        n = len(spans)
        f.write(f"{n} steps: span {sum(spans) / n / 1e6:.3f} ms, some kernel running {sum(unions) / n / 1e6:.3f} ms, "
                f"idle {(sum(spans) - sum(unions)) / n / 1e6:.3f} ms, busy with < 256 workgroups in flight {sum(under) / n / 1e6:.3f} ms\n")
        for q, t in sorted(perq.items()):
            f.write(f"  queue {q}: {t / n / 1e6:.3f} ms of kernel time per step\n")
        st = steps[-pick]
        t0 = st[0][2]
        f.write(f"\nstep {len(steps) - pick}: {len(st)} launches\n{'start us':>9s} {'dur us':>8s} {'end us':>9s}  q {'wgs':>7s}  kernel\n")
        for name, q, s, e, g in st:
            f.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(e - t0) / 1e3:9.1f} {q:2d} {g:7d}  {short(name)}\n")
    print(open(out).read()[:1500])


if __name__ == "__main__":
    main()
