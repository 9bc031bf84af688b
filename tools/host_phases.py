"""Host time per phase of a training step (perf_counter around the phases, no device sync inside):
python tools/host_phases.py [workload] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from graphtrans_amd import engine, ops as gt_ops
from graphtrans_amd.dist import GradSync
from graphtrans_amd.optim import FusedAdamW
from graphtrans_amd.modules import gnn_module

wl = sys.argv[1] if len(sys.argv) > 1 else "molpcba"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
matmul_dtype, dtype = bench.MODES["mixed"]
gt_ops.set_matmul_dtype(matmul_dtype)
per_gpu = {"nci1": 32, "code2-pna": 128}.get(wl, 256)
torch.manual_seed(1234)
args, model, gen, loss_fn, name = bench.build(wl, dtype, dev, per_gpu)
model.train()
sync = GradSync(model.parameters(), world_size=1).attach(model)
optim = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.0)
batches = [bench.attach_sizes(gen(i)).to(dev) for i in range(4)]
T = {}
def tick(k, t0):
    t = time.perf_counter()
    T[k] = T.get(k, 0.0) + (t - t0)
    return t
# finer: wrap a few engine / module functions
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[key] = T.get(key, 0.0) + time.perf_counter() - t0
    setattr(obj, name, g)
wrap(gnn_module, "batch_structure", "  fwd: batch_structure")
from graphtrans_amd.models import gnn_transformer
gnn_transformer.batch_structure = gnn_module.batch_structure
wrap(engine, "forward", "  fwd: engine.forward")
wrap(engine, "eligible", "  fwd: eligible")
from graphtrans_amd import graph
wrap(graph.GraphStructure, "layout", "  fwd: layout")
wrap(engine._FusedModel, "_forward_body", "    engine: _forward_body")
wrap(engine._FusedModel, "_backward_body", "    engine: _backward_body")
for i in range(20 + steps):
    if i == 20:
        torch.cuda.synchronize(); T.clear(); t_all = time.perf_counter()
    b = batches[i % 4]
    b.__dict__.pop("_gt_structure", None)
    t = time.perf_counter()
    sync.zero(); t = tick("zero", t)
    out = model(b); t = tick("forward (model(b))", t)
    loss = loss_fn(out, b); t = tick("loss", t)
    loss.backward(); t = tick("backward", t)
    sync.finish(); t = tick("sync.finish", t)
    optim.step(); t = tick("optim.step", t)
host = time.perf_counter() - t_all
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print(f"{name}: host enqueue {host / steps * 1e3:.3f} ms/step, wall {tot / steps * 1e3:.3f} ms/step")
for k, v in T.items():
    print(f"  {k:34s} {v / steps * 1e6:8.1f} us/step")
