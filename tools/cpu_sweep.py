#!/usr/bin/env python
"""CPU-oracle timing sweep over thread counts: fwd + loss + bwd of oracle/reference_math.py on a
Code2-like sample.  usage: python tools/cpu_sweep.py [graphs] [max seconds per point]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from graphtrans_amd import synth
from oracle import reference_math as rm

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
args = bench.model_args("code2", torch.float32)
torch.manual_seed(0)
from graphtrans_amd.encoders import ASTNodeEncoder
from graphtrans_amd.models.gnn_transformer import GNNTransformer
model = GNNTransformer(5002, ASTNodeEncoder(300, 98, 10030, 20), lambda d: torch.nn.Linear(2, d), args)
b = synth.code2_like(B=graphs, seed=0)
sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
out = {"graphs": graphs, "nodes": int(b.num_nodes), "cpu_count": os.cpu_count(), "points": []}
for th in (16, 8, 32, 4, 64, 1, 128, 256):
    if th > (os.cpu_count() or 1):
        continue
    torch.set_num_threads(th)
    ts = []
    t_start = time.perf_counter()
    for it in range(4):
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        rm.code2_loss(rm.gnn_transformer(sd, args, b, None, True), b.y_arr).backward()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > limit:
            break
    best = min(ts[1:]) if len(ts) > 1 else ts[0]
    out["points"].append({"threads": th, "s_per_step": round(best, 3), "graphs_per_s": round(graphs / best, 2), "iters": len(ts)})
    print(out["points"][-1], flush=True)
print(json.dumps(out))
