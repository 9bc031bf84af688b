#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel stats CSV + per-category table.

usage: python tools/rocpd_summary.py <results.db> <steps> <out_prefix>
Durations in the `top_kernels` view are microseconds... (rocprofv3 --kernel-trace --stats)."""
import csv
import re
import sqlite3
import sys


def category(n):
    ours = re.search(r"\b(k_[a-z_0-9]+)<|\b(k_[a-z_0-9]+)\(", n)
    if "anonymous namespace" in n and ours:
        return "graphtrans_hip:" + (ours.group(1) or ours.group(2))
    if n.startswith("Cijk"):
        return "gemm (hipBLASLt)"
    for key, cat in (("batch_norm", "torch batch_norm"), ("layer_norm", "torch layer_norm"), ("GammaBeta", "torch layer_norm"),
                     ("dropout", "torch dropout"), ("masked_scale", "torch dropout"), ("multi_tensor", "torch fused AdamW"),
                     ("softmax", "torch loss"), ("nll_loss", "torch loss"), ("elementwise", "torch elementwise"),
                     ("reduce_kernel", "torch reduce"), ("rocprim", "torch index/sort (embedding bwd)"),
                     ("sum_and_scatter", "torch index/sort (embedding bwd)"), ("embedding", "torch index/sort (embedding bwd)"),
                     ("index", "torch index/sort (embedding bwd)"), ("rocclr", "runtime fill/copy")):
        if key in n:
            return cat
    return "other"


def main():
    db, steps, prefix = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for n, c, t, a, p in rows:
            w.writerow([n[:200], c, round(t, 1), round(a, 3), round(p, 3)])
    cats = {}
    for n, c, t, a, p in rows:
        d = cats.setdefault(category(n), [0, 0.0])
        d[0] += c
        d[1] += t
    tot = sum(v[1] for v in cats.values())
    with open(prefix + "_summary.txt", "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats summary ({db}), {steps} profiled steps\n")
        f.write(f"GPU kernel time per step: {tot / 1e3 / steps:.3f} ms; launches per step: {sum(v[0] for v in cats.values()) / steps:.0f}\n\n")
        f.write(f"{'category':48s} {'launches/step':>14s} {'ms/step':>9s} {'avg us':>9s} {'share':>7s}\n")
        for k, v in sorted(cats.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k:48s} {v[0] / steps:14.1f} {v[1] / 1e3 / steps:9.3f} {v[1] / max(v[0], 1):9.2f} {100 * v[1] / tot:6.1f}%\n")
    print(open(prefix + "_summary.txt").read())


if __name__ == "__main__":
    main()
