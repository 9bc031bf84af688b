#!/usr/bin/env python
"""Per-stream busy / gap / critical-path table of a rocprofv3 kernel trace (rocpd sqlite) as JSON.

usage: python tools/timeline_json.py <results.db> <out.json> [first_step_from_end=24]

Steps are delimited by the k_adamw launch that ends each one; the last `first_step_from_end` complete steps are
averaged (the timed region of bench.py; the single steps of the idle-device host measurement that follow are cut off by
asking for steps whose span is within 1.5 x the median).  Per step:

* span (first start .. last end), union of busy time, idle time;
* per queue: kernel time, number of launches, sum of the gaps between consecutive launches of that queue;
* the MAIN queue (most kernel time): every gap in front of one of its kernels is attributed to the kernel that follows
  it and classified -- `join` when a kernel of ANOTHER queue ended inside the gap (the main stream waited for an event),
  `dispatch` otherwise (the command processor's launch-to-launch latency, or a host that had not enqueued yet);
* a critical path: from the last kernel of the step backwards, predecessor = the kernel on any queue with the latest end
  that is <= this kernel's start + 1 us.  Kernel time and gap time on that path are summed per kernel family: this is the
  table that ranks what to fuse (a family whose kernels sit on the path costs its duration AND the gap behind each launch).
"""
import json
import re
import sqlite3
import statistics
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"\b(k_[a-z_0-9]+)", n)
    if m:
        return m.group(1)
    if "fillBuffer" in n:
        return "rt_fill"
    if "copyBuffer" in n:
        return "rt_copy"
    return n[:40]


def analyse(st):
    t0, t1 = st[0][2], max(r[3] for r in st)
    ev = sorted([(r[2], 1) for r in st] + [(r[3], -1) for r in st])
    busy = u = 0
    last = t0
    for t, d in ev:
        if busy > 0:
            u += t - last
        last = t
        busy += d
    perq = {}
    for name, q, s, e, g in st:
        d = perq.setdefault(q, {"kernel_us": 0.0, "launches": 0, "gap_us": 0.0, "_last": None})
        if d["_last"] is not None:
            d["gap_us"] += max(0, s - d["_last"]) / 1e3
        d["_last"] = max(e, d["_last"] or 0)
        d["kernel_us"] += (e - s) / 1e3
        d["launches"] += 1
    main = max(perq, key=lambda q: perq[q]["kernel_us"])
    # gaps of the main queue
    gaps = {}
    prev_end = None
    others = sorted((r[3] for r in st if r[1] != main))
    import bisect
    for name, q, s, e, g in st:
        if q != main:
            continue
        if prev_end is not None and s > prev_end:
            i = bisect.bisect_right(others, prev_end)
            kind = "join" if i < len(others) and others[i] <= s else "dispatch"
            d = gaps.setdefault(short(name), {"join_us": 0.0, "dispatch_us": 0.0, "n": 0})
            d[kind + "_us"] += (s - prev_end) / 1e3
            d["n"] += 1
        prev_end = max(e, prev_end or 0)
    # critical path
    by_end = sorted(st, key=lambda r: r[3])
    ends = [r[3] for r in by_end]
    cur = by_end[-1]
    crit = {}
    crit_kernel = crit_gap = 0.0
    n_crit = 0
    while True:
        fam = crit.setdefault(short(cur[0]), {"kernel_us": 0.0, "gap_us": 0.0, "n": 0})
        fam["kernel_us"] += (cur[3] - cur[2]) / 1e3
        fam["n"] += 1
        crit_kernel += (cur[3] - cur[2]) / 1e3
        n_crit += 1
        i = bisect.bisect_right(ends, cur[2] + 1000) - 1
        while i >= 0 and by_end[i] is cur:
            i -= 1
        if i < 0 or by_end[i][2] >= cur[2]:
            # no kernel ended before this one started: start of step (or overlapping launch) -- look for an earlier starter
            cands = [r for r in by_end[:max(i + 1, 0)] if r[2] < cur[2]]
            if not cands:
                break
            nxt = cands[-1]
        else:
            nxt = by_end[i]
        gap = max(0.0, (cur[2] - nxt[3]) / 1e3)
        fam["gap_us"] += gap
        crit_gap += gap
        cur = nxt
        if cur[2] <= t0:
            fam = crit.setdefault(short(cur[0]), {"kernel_us": 0.0, "gap_us": 0.0, "n": 0})
            fam["kernel_us"] += (cur[3] - cur[2]) / 1e3
            fam["n"] += 1
            crit_kernel += (cur[3] - cur[2]) / 1e3
            n_crit += 1
            break
    for d in perq.values():
        d.pop("_last")
    return {"span_us": (t1 - t0) / 1e3, "busy_us": u / 1e3, "launches": len(st), "perq": perq, "main": main, "gaps": gaps,
            "crit": crit, "crit_kernel_us": crit_kernel, "crit_gap_us": crit_gap, "crit_launches": n_crit}


def mean_dicts(ds, keys):
    out = {}
    names = set().union(*[d.keys() for d in ds])
    for n in names:
        out[n] = {k: round(sum(d.get(n, {}).get(k, 0.0) for d in ds) / len(ds), 2) for k in keys}
    return out


def main():
    db, out = sys.argv[1], sys.argv[2]
    last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, queue_id, start, end, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z) "
                            "from kernels order by start"))
    ends = [i for i, r in enumerate(rows) if "k_adamw" in r[0]]
    steps = [rows[ends[i] + 1:ends[i + 1] + 1] for i in range(len(ends) - 1)]
    spans = [max(r[3] for r in s) - s[0][2] for s in steps]
    med = statistics.median(spans)
    # the timed region: back-to-back steps (the start of a step follows the previous step's end closely)
    good = [s for i, s in enumerate(steps) if i > 0 and s[0][2] - max(r[3] for r in steps[i - 1]) < 0.2 * med and spans[i] < 1.5 * med]
    good = good[-last_n:]
    res = [analyse(s) for s in good]
    n = len(res)
    mainq = res[0]["main"]
    qs = sorted(set().union(*[r["perq"].keys() for r in res]))
    doc = {
        "source": db, "steps_averaged": n,
        "span_us": round(sum(r["span_us"] for r in res) / n, 1),
        "some_kernel_running_us": round(sum(r["busy_us"] for r in res) / n, 1),
        "idle_us": round(sum(r["span_us"] - r["busy_us"] for r in res) / n, 1),
        "launches_per_step": round(sum(r["launches"] for r in res) / n, 1),
        "queues": {str(q): {k: round(sum(r["perq"].get(q, {}).get(k, 0) for r in res) / n, 1) for k in ("kernel_us", "launches", "gap_us")} for q in qs},
        "main_queue": str(mainq),
        "main_queue_gaps_by_following_kernel": dict(sorted(mean_dicts([r["gaps"] for r in res], ("join_us", "dispatch_us", "n")).items(),
                                                           key=lambda kv: -(kv[1]["join_us"] + kv[1]["dispatch_us"]))),
        "critical_path": {
            "kernel_us": round(sum(r["crit_kernel_us"] for r in res) / n, 1),
            "gap_us": round(sum(r["crit_gap_us"] for r in res) / n, 1),
            "launches": round(sum(r["crit_launches"] for r in res) / n, 1),
            "by_family": dict(sorted(mean_dicts([r["crit"] for r in res], ("kernel_us", "gap_us", "n")).items(),
                                     key=lambda kv: -(kv[1]["kernel_us"] + kv[1]["gap_us"]))),
        },
    }
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: doc[k] for k in ("steps_averaged", "span_us", "some_kernel_running_us", "idle_us", "launches_per_step", "queues")}))
    cp = doc["critical_path"]
    print(f"critical path: {cp['launches']} launches, kernel {cp['kernel_us']} us + gaps {cp['gap_us']} us")
    for k, v in list(cp["by_family"].items())[:40]:
        print(f"  {k:28s} n {v['n']:5.1f}  kernel {v['kernel_us']:8.1f}  gap {v['gap_us']:7.1f}")


if __name__ == "__main__":
    main()
