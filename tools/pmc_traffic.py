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

# bench.py kernel-report key -> (regexes over the demangled kernel names it launches, regex of the kernel whose call count
# = the entry point's call count).  k_split_reduce (the fixed-order reduce of the dW partials) is shared by the fp32 and
# bf16 GEMM paths and is reported on its own (r6: bench.py's brackets hold the GEMM kernel alone as well).
BF16 = r"unsigned short"
ENTRY = {
    # GEMM kernels one by one (bench.py kernel_report keys = the C-side kernel-level profiler names, GT_PROF_GEMM_KERNEL)
    "k_lin3[fwd]": ((r"k_lin3<[^>]*, false, (?:false|true)>",), r"k_lin3<[^>]*, false, (?:false|true)>"),
    "k_lin3[dx]": ((r"k_lin3<[^>]*, true, false>",), r"k_lin3<[^>]*, true, false>"),
    # k_lin3r (rows straight into fragments, r5): forward and dX are ONE instantiation -- their launches are pooled, both keys get the average
    "k_lin3r": ((r"k_lin3r<",), r"k_lin3r<"),        # (bench.py's pooled family = rocprofv3's one row)
    "k_lin3r[fwd]": ((r"k_lin3r<",), r"k_lin3r<"),
    "k_lin3r[dx]": ((r"k_lin3r<",), r"k_lin3r<"),
    "k_lin3r_dw": ((r"k_lin3r_dw<",), r"k_lin3r_dw<"),
    "k_lin32[fwd]": ((r"k_lin32<[^>]*?, \d+, false,",), r"k_lin32<[^>]*?, \d+, false,"),
    "k_lin32[dx]": ((r"k_lin32<[^>]*?, \d+, true,", r"k_transpose32"), r"k_lin32<[^>]*?, \d+, true,"),
    # weight-stationary encoder GEMMs: forward and dX share instantiations (k_lin1<KS, NTW, LN>), so the plain ones are pooled
    "k_lin1[fwd+ln][bf16]": ((r"k_lin1<\d+, \d+, (?:true|1)>",), r"k_lin1<\d+, \d+, (?:true|1)>"),
    "k_lin1[bf16]": ((r"k_lin1<\d+, \d+, (?:false|0)>",), r"k_lin1<\d+, \d+, (?:false|0)>"),   # (bench.py's pooled family)
    "k_lin1[fwd|dx][bf16]": ((r"k_lin1<\d+, \d+, (?:false|0)>",), r"k_lin1<\d+, \d+, (?:false|0)>"),
    "k_lin3_dw": ((r"k_lin3_dw<",), r"k_lin3_dw<"),     # (its k_split_reduce launches are shared with the other dW kernels: reported on their own)
    "k_lin32_dw": ((r"k_lin32_dw<",), r"k_lin32_dw<"),   # (its k_split_reduce launches are shared with the bf16 path: reported on their own)
    "k_linear_fwd[bf16]": ((r"k_linear_fwd<[^>]*" + BF16 + r", \d+>",), r"k_linear_fwd<[^>]*" + BF16 + r", \d+>"),
    "k_linear_dx[bf16]": ((r"k_linear_dx<[^>]*" + BF16 + r", \d+>",), r"k_linear_dx<[^>]*" + BF16 + r", \d+>"),
    "k_linear_dw[bf16]": ((r"k_linear_dw<[^>]*" + BF16 + r">",), r"k_linear_dw<[^>]*" + BF16 + r">"),
    # the round-1 tiled kernels in exact fp32: what PNA's grouped tower GEMMs still run on
    "k_linear_fwd[fp32]": ((r"k_linear_fwd<[^>]*float, \d+>",), r"k_linear_fwd<[^>]*float, \d+>"),
    "k_linear_dx[fp32]": ((r"k_linear_dx<[^>]*float, \d+>",), r"k_linear_dx<[^>]*float, \d+>"),
    "k_linear_dw[fp32]": ((r"k_linear_dw<[^>]*float>",), r"k_linear_dw<[^>]*float>"),
    "k_dw16[bf16]": ((r"k_dw16\(",), r"k_dw16\("),   # LDS-DMA ring dW of the encoder linears (its k_split_reduce: reported on its own)
    "k_split_reduce": ((r"k_split_reduce",), r"k_split_reduce"),
    "gt_aggregate_fwd": ((r"k_aggw?_fwd<",), r"k_aggw?_fwd<"),
    "gt_aggregate_bwd": ((r"k_aggw?_bwd<",), r"k_aggw?_bwd<"),   # (the gather kernel: its parameter-partials reduce, k_agg_reduce, runs on the overlap stream)
    "k_agg_reduce": ((r"k_agg_reduce",), r"k_agg_reduce"),
    "gt_attn_fwd": ((r"k_attn_fwd<",), r"k_attn_fwd<"),
    "gt_attn_bwd": ((r"k_attn_bwd_dq<", r"k_attn_bwd_dkv<"), r"k_attn_bwd_dq<"),
}


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    agg = {}
    for n, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if c != counter:
            continue
        full = re.sub(r"\(anonymous namespace\)::", "", n)[:120]
        if not re.search(r"\bk_[a-z_0-9]+", full):
            continue
        a = agg.setdefault(full, [0, 0.0])
        a[0] += 1
        a[1] += v
    return agg


def per_kernel_raw(path, counter):
    """the same from a *_pmc_traffic_raw.txt written by an earlier run (name, counter, calls=, avg=)"""
    agg = {}
    for line in open(path):
        m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+calls=\s*(\d+)\s+avg=\s*([0-9.]+)", line)
        if m and m.group(2) == counter:
            agg[m.group(1).strip()] = [int(m.group(3)), int(m.group(3)) * float(m.group(4))]
    return agg


def traffic_of(fetch, write):
    traffic = {}
    for ep, (patterns, counted) in ENTRY.items():
        calls = sum(c for full, (c, _t) in fetch.items() if re.search(counted, full))
        if not calls:
            continue
        kb = sum(2.0 * t for full, (_c, t) in fetch.items() if any(re.search(p, full) for p in patterns)) + \
            sum(t for full, (_c, t) in write.items() if any(re.search(p, full) for p in patterns))
        traffic[ep] = int(kb * 1024 / calls)
    return traffic


def main():
    if sys.argv[1] == "--from-raw":   # python tools/pmc_traffic.py --from-raw <raw.txt> <existing.json>: re-key an earlier run
        raw, js = sys.argv[2:4]
        d = json.load(open(js))
        d["traffic"] = traffic_of(per_kernel_raw(raw, "FETCH_SIZE"), per_kernel_raw(raw, "WRITE_SIZE"))
        json.dump(d, open(js, "w"), indent=1)
        print(json.dumps(d["traffic"]))
        return
    fdb, wdb, workload, mode, per_gpu, prefix = sys.argv[1:7]
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench   # build_id(): the sources these counters were measured on; bench.py attaches the file to that build only
    fetch, write = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    with open(prefix + "_pmc_traffic_raw.txt", "w") as f:
        for name, agg in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            for full, (calls, tot) in sorted(agg.items()):
                f.write(f"{full:92s} {name:12s} calls={calls:5d} avg={tot / calls:14.1f}\n")
    traffic = traffic_of(fetch, write)
    out = {"workload": workload, "mode": mode, "graphs_per_gpu": int(per_gpu), "build_id": bench.build_id(),
           "unit": "bytes per entry-point call (2*FETCH_SIZE + WRITE_SIZE, KB counters, rocprofv3 --pmc, one counter per pass)",
           "traffic": traffic}
    json.dump(out, open(prefix + "_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(traffic))


if __name__ == "__main__":
    main()
