"""Host time per phase of a training step (perf_counter around the phases, no device sync inside):
python tools/host_phases.py [workload] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from graphtrans_amd import engine, ops as gt_ops
from graphtrans_amd.dist import GradSync
from graphtrans_amd.optim import FusedAdamW

wl = sys.argv[1] if len(sys.argv) > 1 else "molpcba"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
matmul_dtype, dtype = bench.MODES["mixed"]
gt_ops.set_matmul_dtype(matmul_dtype)
per_gpu = int(sys.argv[3]) if len(sys.argv) > 3 else {"nci1": 32, "code2-pna": 128}.get(wl, 256)
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
wrap(engine, "forward", "  fwd: engine.forward")
wrap(engine, "eligible", "  fwd: eligible")
L = engine._lib.lib()
for fn in ("gt_model_prepare", "gt_model_forward", "gt_model_backward"):   # time inside the C driver (launches + event traffic)
    f0 = getattr(L, fn)
    def mk(f0=f0, fn=fn):
        def g(*a):
            t0 = time.perf_counter()
            try:
                return f0(*a)
            finally:
                T["      C: " + fn] = T.get("      C: " + fn, 0.0) + time.perf_counter() - t0
        return g
    setattr(L, fn, mk())
prof = None
import gc
if os.environ.get('GT_GC') == 'off':
    gc.disable()
elif os.environ.get('GT_GC') == 'freeze':
    gc.collect(); gc.freeze()
elif os.environ.get('GT_GC') == 'debug':
    gc.callbacks.append(lambda ph, info: T.__setitem__('gc gen%d' % info['generation'], T.get('gc gen%d' % info['generation'], 0.0) + (1e-6 if ph == 'start' else 0.0)))
for i in range(20 + steps):
    if i == 20:
        torch.cuda.synchronize(); T.clear(); t_all = time.perf_counter()
        if os.environ.get("GT_CPROFILE"):
            import cProfile
            prof = cProfile.Profile(); prof.enable()
    b = batches[i % 4]
    b.__dict__.pop("_gt_structure", None)
    t = time.perf_counter()
    sync.zero(); t = tick("zero", t)
    out = model(b); t = tick("forward (model(b))", t)
    loss = loss_fn(out, b); t = tick("loss", t)
    loss.backward(); t = tick("backward", t)
    sync.finish(); t = tick("sync.finish", t)
    optim.step(); t = tick("optim.step", t)
if prof is not None:
    prof.disable()
host = time.perf_counter() - t_all
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print(f"{name}: host enqueue {host / steps * 1e3:.3f} ms/step, wall {tot / steps * 1e3:.3f} ms/step")
for k, v in T.items():
    print(f"  {k:34s} {v / steps * 1e6:8.1f} us/step")
if prof is not None:
    import pstats, io
    for key in ("tottime", "cumulative"):
        b_ = io.StringIO()
        pstats.Stats(prof, stream=b_).sort_stats(key).print_stats(28)
        print(b_.getvalue()[:6000])
