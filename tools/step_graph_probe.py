#!/usr/bin/env python
"""Does the whole training step (graph_prep -> fused forward -> loss -> fused backward -> AdamW, three streams) capture into
a hipGraph, and what does replay cost?  usage: python tools/step_graph_probe.py [workload=code2] [mode=mixed] [steps=50]
(measurement probe: dropout seeds and the AdamW step count are frozen in the captured graphs)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from graphtrans_amd import ops as gt_ops
from graphtrans_amd.dist import GradSync
from graphtrans_amd.graph import GraphStructure
from graphtrans_amd.modules.gnn_module import batch_structure
from graphtrans_amd.optim import FusedAdamW

WL = sys.argv[1] if len(sys.argv) > 1 else "code2"
MODE = sys.argv[2] if len(sys.argv) > 2 else "mixed"
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 50
device = torch.device("cuda:0")
torch.cuda.set_device(device)
matmul, tok = bench.MODES[MODE]
gt_ops.set_matmul_dtype(matmul)
per_gpu = {"nci1": 32, "code2-pna": 128}.get(WL, 256)
torch.manual_seed(1234)
args, model, gen, loss_fn, _ = bench.build(WL, tok, device, per_gpu)
model.train()
sync = GradSync(model.parameters(), world_size=1).attach(model)
optim = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.0)
batches = [bench.attach_sizes(gen(i)).to(device) for i in range(4)]
enc = model.transformer_encoder
key = ("packed", int(enc.max_input_len), enc.cls_embedding is not None)
lays = []
for b in batches:   # host-built exact layouts, uploaded once and kept alive beside the graphs
    gs0 = GraphStructure.build(b.edge_index, b.batch, num_graphs=b.num_graphs, sizes=b._sizes)
    lays.append(gs0.layout(*key))


def step(i, inject=False):
    b = batches[i % 4]
    b.__dict__.pop("_gt_structure", None)
    gs = batch_structure(b)
    if inject:
        gs._layouts[key] = lays[i % 4]
    sync.zero()
    out = model(b)
    loss = loss_fn(out, b)
    loss.backward()
    sync.finish()
    optim.step()
    return loss


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n


for i in range(12):
    step(i, True)
h, t = timeit(lambda i: step(i, True), STEPS)
print(f"eager : {t:.3f} ms/step  (host enqueue {h:.3f} ms)  {per_gpu / t * 1e3:.0f} graphs/s", flush=True)

graphs, losses = [], []
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(4):
        step(i, True)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
ok = True
try:
    for i in range(4):
        g = torch.cuda.CUDAGraph()
        t0 = time.perf_counter()
        with torch.cuda.graph(g):
            l = step(i, True)
        torch.cuda.synchronize()
        print(f"captured batch {i} in {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
        graphs.append(g)
        losses.append(l)
except Exception as e:
    ok = False
    print("CAPTURE FAILED:", repr(e), flush=True)
if ok:
    for i in range(8):
        graphs[i % 4].replay()
    torch.cuda.synchronize()
    print("replay losses:", [round(float(l), 5) for l in losses], flush=True)
    h, t = timeit(lambda i: graphs[i % 4].replay(), STEPS)
    print(f"replay: {t:.3f} ms/step  (host enqueue {h:.3f} ms)  {per_gpu / t * 1e3:.0f} graphs/s", flush=True)
    h, t = timeit(lambda i: graphs[i % 4].replay(), STEPS * 4)
    print(f"replay: {t:.3f} ms/step  (host enqueue {h:.3f} ms)  {per_gpu / t * 1e3:.0f} graphs/s", flush=True)
