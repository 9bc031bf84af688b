#!/usr/bin/env python
"""gt_aggregate_fwd / _bwd alone on one batch of BASELINE configs[4] (256 x G(512, 8/511), D = 256): bench.py's aggregate_stress
leg as its own command, so that a rocprofv3 --pmc pass over it yields the HBM traffic of exactly these launches.
usage: python tools/agg_stress.py            -> prints the report JSON
       python tools/agg_stress.py --pmc-json <fetch.db> <write.db> <out.json>   -> bytes per launch from two PMC passes"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "--pmc-json":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import re

    import bench
    import pmc_traffic as pt
    fetch, write = pt.per_kernel(sys.argv[2], "FETCH_SIZE"), pt.per_kernel(sys.argv[3], "WRITE_SIZE")
    out = {"workload": "aggregate_stress", "build_id": bench.build_id(),
           "unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, KB counters, separate rocprofv3 --pmc passes)", "traffic": {}}
    for key, pat in (("gt_aggregate_fwd", r"k_aggw?_fwd<"), ("gt_aggregate_bwd", r"k_aggw?_bwd<")):
        calls = sum(c for n, (c, _t) in fetch.items() if re.search(pat, n))
        if calls:
            kb = sum(2.0 * t for n, (_c, t) in fetch.items() if re.search(pat, n)) + sum(t for n, (_c, t) in write.items() if re.search(pat, n))
            out["traffic"][key] = int(kb * 1024 / calls)
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out))
    sys.exit(0)

import torch

import bench

torch.cuda.set_device(0)
print(json.dumps(bench.aggregate_stress_report(torch.device("cuda", 0))))
