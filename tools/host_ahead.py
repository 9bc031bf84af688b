#!/usr/bin/env python
"""Is the training loop bound by the host's enqueue time?  UNPROFILED: per step, the host time at which step i's first launch is
enqueued and the device time at which it starts (an event recorded there); lead_i = device - host, both measured from step 0.  A lead
that grows and saturates (the queue's back-pressure) = the GPU never waits for the host; a lead near zero = enqueue-bound.
usage: python tools/host_ahead.py [workload=code2] [steps=60] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from graphtrans_amd import ops
from graphtrans_amd.dist import GradSync
from graphtrans_amd.optim import FusedAdamW

wl = sys.argv[1] if len(sys.argv) > 1 else "code2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
per_gpu = int(sys.argv[3]) if len(sys.argv) > 3 else (32 if wl == "nci1" else 256)
dev = torch.device("cuda:0")
ops.set_matmul_dtype(torch.float32)
torch.manual_seed(0)
args, model, gen, loss_fn, _ = bench.build(wl, torch.bfloat16, dev, per_gpu)
model.train()
sync = GradSync(model.parameters(), world_size=1).attach(model)
optim = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.0)
batches = [bench.attach_sizes(gen(i)).to(dev) for i in range(4)]


def step(i):
    b = batches[i % 4]
    b.__dict__.pop("_gt_structure", None)
    sync.zero()
    loss = loss_fn(model(b), b)
    loss.backward()
    sync.finish()
    optim.step()


for i in range(15):
    step(i)
torch.cuda.synchronize()
cur = torch.cuda.current_stream(0)
marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host = []
for i in range(steps):
    host.append(time.perf_counter())
    marks[i].record(cur)
    step(i)
host.append(time.perf_counter())
marks[steps].record(cur)
torch.cuda.synchronize()
t_end = time.perf_counter()
print(f"{wl} b{per_gpu}: host enqueued {steps} steps in {1e3 * (host[-1] - host[0]):.1f} ms, device ran them in {marks[0].elapsed_time(marks[steps]):.1f} ms")
for i in (0, 1, 2, 3, 5, 8, 12, 20, 30, 40, steps - 1, steps):
    if i <= steps:
        print(f"  step {i:3d}: host at {1e3 * (host[i] - host[0]):8.2f} ms, device at {marks[0].elapsed_time(marks[i]):8.2f} ms, lead {marks[0].elapsed_time(marks[i]) - 1e3 * (host[i] - host[0]):7.2f} ms")
