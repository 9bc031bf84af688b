#!/usr/bin/env python
"""Where does forward noise come from?  Runs the C5 encoder stack op by op (HIP exact-fp32 ops vs torch float64 vs torch
fp32, all on the GPU) and reports, per layer, the relative error of the linear1 pre-activation and the number of ReLU
units whose sign differs from the float64 run.  Diagnostic tool (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.nn.functional as F

from graphtrans_amd import ops
from test_hip_attention import make_layout

DEV = "cuda:0"
torch.manual_seed(2)
d, nhead, ffn, L = 256, 4, 1024, 4
lens = [513, 513, 301, 66, 513, 2]
lay = make_layout("packed", lens)
rows = lay.rows
P = []
for _ in range(L):
    P.append(dict(in_w=torch.randn(3 * d, d) / d ** 0.5, in_b=torch.randn(3 * d) * 0.1, out_w=torch.randn(d, d) / d ** 0.5,
                  out_b=torch.randn(d) * 0.1, w1=torch.randn(ffn, d) / d ** 0.5, b1=torch.randn(ffn) * 0.1,
                  w2=torch.randn(d, ffn) / ffn ** 0.5, b2=torch.randn(d) * 0.1, g1=torch.rand(d) + 0.5, be1=torch.randn(d) * 0.1,
                  g2=torch.rand(d) + 0.5, be2=torch.randn(d) * 0.1))
x0 = torch.randn(rows, d)


def torch_attn(qkv, dtype):
    hd = d // nhead
    outs = []
    for (row0, npos, _, _) in lay.desc_cpu:
        x = qkv[row0:row0 + npos]
        q, k, v = [t.view(npos, nhead, hd).transpose(0, 1) for t in x.split(d, dim=1)]
        p = torch.softmax((q @ k.transpose(1, 2)) * hd ** -0.5, -1)
        outs.append((p @ v).transpose(0, 1).reshape(npos, d))
    return torch.cat(outs)


def run(kind):
    dt = torch.float64 if kind == "f64" else torch.float32
    x = x0.to(DEV).to(dt)
    pre = []
    for p in P:
        q = {k: v.to(DEV).to(dt) for k, v in p.items()}
        if kind == "hip":
            qkv = ops.linear(x, q["in_w"], q["in_b"])
            ctx = ops.attention(qkv, lay, nhead)
            a = ops.linear(ctx, q["out_w"], q["out_b"])
            x = ops.layer_norm(a, q["g1"], q["be1"], 1e-5, resid=x)
            z = ops.linear(x, q["w1"], q["b1"])
            f = ops.linear(torch.relu(z), q["w2"], q["b2"])
            x = ops.layer_norm(f, q["g2"], q["be2"], 1e-5, resid=x)
        else:
            qkv = F.linear(x, q["in_w"], q["in_b"])
            ctx = torch_attn(qkv, dt)
            a = F.linear(ctx, q["out_w"], q["out_b"])
            x = F.layer_norm(x + a, (d,), q["g1"], q["be1"], 1e-5)
            z = F.linear(x, q["w1"], q["b1"])
            f = F.linear(torch.relu(z), q["w2"], q["b2"])
            x = F.layer_norm(x + f, (d,), q["g2"], q["be2"], 1e-5)
        pre.append(z.double())
    return pre


ops.set_matmul_dtype(torch.float32)
r64, r32, hip = run("f64"), run("f32"), run("hip")
for l in range(L):
    for name, r in (("torch-fp32", r32[l]), ("hip", hip[l])):
        err = (r - r64[l]).abs()
        flips = int(((r > 0) != (r64[l] > 0)).sum())
        print(f"layer {l} {name:10s} pre-activation rel-L2 {float(err.norm() / r64[l].norm()):.2e} max abs {float(err.max()):.2e} "
              f"flipped units {flips} of {r.numel()}")
