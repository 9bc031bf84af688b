#!/usr/bin/env python
"""Host enqueue time per phase of bench.py's step (perf_counter, no device sync inside the loop).
usage: python tools/host_phases.py [graphs_per_gpu=8] [steps=200] [workload=code2]   -- a tiny batch keeps the GPU
ahead of the host, so the numbers are pure host cost."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from graphtrans_amd import ops as gt_ops
from graphtrans_amd.dist import GradSync

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
WL = sys.argv[3] if len(sys.argv) > 3 else "code2"
device = torch.device("cuda:0")
torch.cuda.set_device(device)
gt_ops.set_matmul_dtype(torch.bfloat16)
torch.manual_seed(1234)
args, model, gen, loss_fn, _ = bench.build(WL, torch.bfloat16, device, B)
model.train()
sync = GradSync(model.parameters(), world_size=1)
from graphtrans_amd.optim import FusedAdamW
optim = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.0)
batches = [bench.attach_sizes(gen(i)).to(device) for i in range(4)]
acc = {}
scratch = torch.empty(16, device=device)


def lap(name, t):
    now = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + now - t
    return now


for i in range(steps + 20):
    if i == 20:
        torch.cuda.synchronize()
        acc.clear()
        t_all = time.perf_counter()
    b = batches[i % 4]
    b.__dict__.pop("_gt_structure", None)
    t = time.perf_counter()
    sync.zero(); t = lap("zero", t)
    from graphtrans_amd.modules.gnn_module import batch_structure
    batch_structure(b); t = lap("graph_prep", t)
    out = model(b); t = lap("forward", t)
    loss = loss_fn(out, b); t = lap("loss", t)
    loss.backward(); t = lap("backward", t)
    sync.finish(); t = lap("grad_sync", t)
    if os.environ.get("GT_PROBE"):
        gl = [p.grad for p in sync.params]; t = lap("probe_grad_access", t)
        scratch.zero_(); t = lap("probe_tiny_launch", t)
        scratch.zero_(); t = lap("probe_tiny_launch2", t)
    optim.step(); t = lap("adamw", t)
total = time.perf_counter() - t_all
torch.cuda.synchronize()
print(f"B={B}: host {1e3 * total / steps:.3f} ms/step")
for k, v in acc.items():
    print(f"  {k:12s} {1e3 * v / steps:7.3f} ms")
