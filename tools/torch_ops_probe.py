"""Which torch ops still launch kernels inside a fused step (python tools/torch_ops_probe.py): aten op names with device time."""
import sys
sys.path.insert(0, ".")
import torch
import bench
from graphtrans_amd import ops, optim as gopt

dev = torch.device("cuda:0")
ops.set_matmul_dtype(torch.float32)
torch.manual_seed(0)
args, model, gen, loss_fn, _ = bench.build("code2", torch.bfloat16, dev, 256)
model.train()
opt = gopt.FusedAdamW(model.parameters(), lr=1e-4)
bs = [bench.attach_sizes(gen(i)).to(dev) for i in range(2)]


def step(i):
    b = bs[i % 2]
    b.__dict__.pop("_gt_structure", None)
    opt.zero_grad(set_to_none=True)
    loss = loss_fn(model(b), b)
    loss.backward()
    opt.step()


for i in range(5):
    step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=4) if e.key.startswith("aten::") and getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)) > 0]
for e in sorted(rows, key=lambda e: -getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))):
    t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    print(f"{e.key:28s} calls {e.count:3d} device us {t:8.1f}")
    for s in (e.stack or [])[:4]:
        print("      ", s)
