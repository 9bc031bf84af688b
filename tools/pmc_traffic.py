#!/usr/bin/env python
"""HBM-side traffic per entry-point call from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: python tools/pmc_traffic.py <fetch.db> <write.db> <workload> <mode> <graphs_per_gpu> <out_prefix>
Writes <out_prefix>_pmc_traffic.json (read by bench.py for roofline.traffic) and <out_prefix>_pmc_traffic_raw.txt.
Counter handling as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KB, and
FETCH_SIZE under-reports wide coalesced reads by a factor 2 -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Entry point = the kernels it launches: traffic per call = sum over those kernels / number of entry-point calls."""
import json
import re
import sqlite3
import sys

ENTRY = {   # entry point -> (kernel name prefixes, the kernel whose call count = the entry point's call count)
    "gt_linear_fwd": (("k_linear_fwd",), "k_linear_fwd"),
    "gt_linear_bwd": (("k_linear_dx", "k_linear_dw", "k_split_reduce"), "k_linear_dw"),
    "gt_aggregate_fwd": (("k_agg_fwd",), "k_agg_fwd"),
    "gt_aggregate_bwd": (("k_agg_bwd", "k_agg_reduce"), "k_agg_bwd"),
    "gt_attn_fwd": (("k_attn_fwd",), "k_attn_fwd"),
    "gt_attn_bwd": (("k_attn_bwd_dq", "k_attn_bwd_dkv", "k_attn_bwd_prep"), "k_attn_bwd_dq"),
}


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    agg = {}
    for n, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if c != counter:
            continue
        m = re.search(r"\b(k_[a-z_0-9]+)", n)
        if not m:
            continue
        a = agg.setdefault((m.group(1), re.sub(r"\(anonymous namespace\)::", "", n)[:90]), [0, 0.0])
        a[0] += 1
        a[1] += v
    return agg


def main():
    fdb, wdb, workload, mode, per_gpu, prefix = sys.argv[1:7]
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench   # build_id(): the sources these counters were measured on; bench.py attaches the file to that build only
    fetch, write = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    with open(prefix + "_pmc_traffic_raw.txt", "w") as f:
        for name, agg in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            for (k, full), (calls, tot) in sorted(agg.items(), key=lambda kv: kv[0][1]):
                f.write(f"{full:92s} {name:12s} calls={calls:5d} avg={tot / calls:14.1f}\n")
    traffic = {}
    for ep, (prefixes, counted) in ENTRY.items():
        calls = sum(c for (k, _), (c, _t) in fetch.items() if k == counted)
        if not calls:
            continue
        kb = sum(2.0 * t for (k, _), (_c, t) in fetch.items() if k in prefixes) + \
            sum(t for (k, _), (_c, t) in write.items() if k in prefixes)
        traffic[ep] = int(kb * 1024 / calls)
    out = {"workload": workload, "mode": mode, "graphs_per_gpu": int(per_gpu), "build_id": bench.build_id(),
           "unit": "bytes per entry-point call (2*FETCH_SIZE + WRITE_SIZE, KB counters, rocprofv3 --pmc, one counter per pass)",
           "traffic": traffic}
    json.dump(out, open(prefix + "_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(traffic))


if __name__ == "__main__":
    main()
