"""cProfile of the fused engine's forward / backward bodies (the backward runs on autograd's thread: profiled from inside)."""
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from graphtrans_amd import engine, ops as gt_ops
from graphtrans_amd.dist import GradSync
from graphtrans_amd.optim import FusedAdamW
wl = sys.argv[1] if len(sys.argv) > 1 else "code2"
dev = torch.device("cuda:0")
matmul_dtype, dtype = bench.MODES["mixed"]
gt_ops.set_matmul_dtype(matmul_dtype)
torch.manual_seed(1234)
args, model, gen, loss_fn, name = bench.build(wl, dtype, dev, 256)
model.train()
sync = GradSync(model.parameters(), world_size=1).attach(model)
optim = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.0)
batches = [bench.attach_sizes(gen(i)).to(dev) for i in range(4)]
profs = {"fwd": cProfile.Profile(), "bwd": cProfile.Profile()}
on = [False]
def wrap(name, key):
    f = getattr(engine._FusedModel, name)
    def g(*a, **k):
        if not on[0]:
            return f(*a, **k)
        profs[key].enable()
        try:
            return f(*a, **k)
        finally:
            profs[key].disable()
    setattr(engine._FusedModel, name, staticmethod(g))
wrap("_forward_body", "fwd")
wrap("_backward_body", "bwd")
N = 60
for i in range(20 + N):
    on[0] = i >= 20
    b = batches[i % 4]
    b.__dict__.pop("_gt_structure", None)
    sync.zero(); out = model(b); loss = loss_fn(out, b); loss.backward(); sync.finish(); optim.step()
torch.cuda.synchronize()
for key in ("fwd", "bwd"):
    s = io.StringIO()
    st = pstats.Stats(profs[key], stream=s)
    st.sort_stats("tottime").print_stats(22)
    txt = s.getvalue()
    print(f"==== {key}: per step = totals / {N}")
    print(txt[:4200])
