#!/usr/bin/env python
"""Accuracy of each exact-fp32 HIP op at Code2 b256 dims against a float64 evaluation of the same math (GPU torch),
with torch's own fp32 evaluation beside it as the yardstick.  Prints relative-L2 and scale-relative max errors of the
output and of every gradient.  (GPU box; diagnostic tool, not a test.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from graphtrans_amd import ops, synth
from graphtrans_amd.graph import GraphStructure

DEV = "cuda:0"
torch.manual_seed(0)


def errs(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max())


ROWS = []   # (case, tensor, hip rel-L2, torch-fp32 rel-L2): read by tests/test_hip_fp32_accuracy.py


def report(name, hip, t32, r64):
    for label in r64:
        eh, et = errs(hip[label], r64[label]), errs(t32[label], r64[label])
        ROWS.append((name, label, eh[0], et[0]))
        print(f"{name:18s} {label:8s} hip l2 {eh[0]:.2e} max {eh[1]:.2e} | torch-fp32 l2 {et[0]:.2e} max {et[1]:.2e} | ratio {eh[0] / max(et[0], 1e-30):.1f}")


def run(fn, inputs, w, dtype):
    xs = {k: (v.detach().to(dtype).requires_grad_(True) if v.is_floating_point() else v) for k, v in inputs.items()}
    out = fn(**xs)
    (out * w.to(out.dtype)).sum().backward()
    res = {"out": out.detach()}
    res.update({"d" + k: v.grad for k, v in xs.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None})
    return res


def linear_case(M, N, K):
    x, W, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / K ** 0.5, torch.randn(N, device=DEV)
    w = torch.randn(M, N, device=DEV)
    ops.set_matmul_dtype(torch.float32)
    hip = run(lambda x, W, b: ops.linear(x, W, b), dict(x=x, W=W, b=b), w, torch.float32)
    t32 = run(lambda x, W, b: F.linear(x, W, b), dict(x=x, W=W, b=b), w, torch.float32)
    r64 = run(lambda x, W, b: F.linear(x, W, b), dict(x=x, W=W, b=b), w, torch.float64)
    report(f"linear {M}x{N}x{K}", hip, t32, r64)


def ln_case(rows, d):
    x, r, g, b = torch.randn(rows, d, device=DEV), torch.randn(rows, d, device=DEV), torch.rand(d, device=DEV) + 0.5, torch.randn(d, device=DEV)
    w = torch.randn(rows, d, device=DEV)
    hip = run(lambda x, r, g, b: ops.layer_norm(x, g, b, 1e-5, resid=r), dict(x=x, r=r, g=g, b=b), w, torch.float32)
    f = lambda x, r, g, b: F.layer_norm(x + r, (d,), g, b, 1e-5)
    report(f"layernorm {rows}x{d}", hip, run(f, dict(x=x, r=r, g=g, b=b), w, torch.float32), run(f, dict(x=x, r=r, g=g, b=b), w, torch.float64))


def bn_case(N, D, relu):
    x = torch.randn(N, D, device=DEV) * 2 + torch.randn(D, device=DEV) * 3   # per-column offsets like real activations
    g, b = torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV)
    w = torch.randn(N, D, device=DEV)
    rm, rv, nbt = torch.zeros(D, device=DEV), torch.ones(D, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    hip = run(lambda x, g, b: ops.batch_norm(x, g, b, rm.clone(), rv.clone(), nbt.clone(), 0.1, 1e-5, True, relu=relu), dict(x=x, g=g, b=b), w, torch.float32)

    def f(x, g, b):
        y = F.batch_norm(x, None, None, g, b, True, 0.1, 1e-5)
        return torch.relu(y) if relu else y
    report(f"batchnorm{'+relu' if relu else ''} {N}x{D}", hip, run(f, dict(x=x, g=g, b=b), w, torch.float32), run(f, dict(x=x, g=g, b=b), w, torch.float64))


def attn_case(lens, d, nhead):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_hip_attention import make_layout
    lay = make_layout("packed", lens)
    qkv = torch.randn(lay.rows, 3 * d, device=DEV)
    w = torch.randn(lay.rows, d, device=DEV)
    hd = d // nhead
    hip = run(lambda qkv: ops.attention(qkv, lay, nhead), dict(qkv=qkv), w, torch.float32)

    def f(qkv):
        outs = []
        for (row0, npos, _, _) in lay.desc_cpu:
            x = qkv[row0:row0 + npos]
            q, k, v = [t.view(npos, nhead, hd).transpose(0, 1) for t in x.split(d, dim=1)]
            p = torch.softmax((q @ k.transpose(1, 2)) * hd ** -0.5, -1)
            outs.append((p @ v).transpose(0, 1).reshape(npos, d))
        return torch.cat(outs)
    report(f"attention d{d} h{nhead}", hip, run(f, dict(qkv=qkv), w, torch.float32), run(f, dict(qkv=qkv), w, torch.float64))


def agg_case(B, D):
    from graphtrans_amd.modules.conv import edge_spec
    b = synth.code2_like(B=B, seed=3).to(DEV)
    N = b.num_nodes
    gs = GraphStructure.build(b.edge_index, b.batch, num_graphs=b.num_graphs)
    h = torch.randn(N, D, device=DEV)
    root = torch.randn(1, D, device=DEV) * 0.3
    We, be = torch.randn(D, 2, device=DEV) * 0.5, torch.randn(D, device=DEV) * 0.3
    w = torch.randn(N, D, device=DEV)

    def hipf(h, root, We, be):
        enc = torch.nn.Linear(2, D).to(DEV)
        enc.weight, enc.bias = torch.nn.Parameter(We.detach()), torch.nn.Parameter(be.detach())
        hipf.enc = enc
        return ops.aggregate(h, gs, "gcn", root, edge_spec(enc, b.edge_attr, D))
    xs = dict(h=h.clone().requires_grad_(True), root=root.clone().requires_grad_(True))
    out = hipf(xs["h"], xs["root"], We, be)
    (out * w).sum().backward()
    hip = dict(out=out.detach(), dh=xs["h"].grad, droot=xs["root"].grad, dWe=hipf.enc.weight.grad, dbe=hipf.enc.bias.grad)

    def f(h, root, We, be):
        row, col = b.edge_index[0], b.edge_index[1]
        deg = torch.zeros(N, dtype=h.dtype, device=DEV).index_add_(0, row, torch.ones(row.numel(), dtype=h.dtype, device=DEV)) + 1
        dis = deg.pow(-0.5)
        e = b.edge_attr.to(h.dtype) @ We.t() + be
        msg = (dis[row] * dis[col]).unsqueeze(1) * torch.relu(h[row] + e)
        return torch.zeros_like(h).index_add_(0, col, msg) + torch.relu(h + root) / deg.unsqueeze(1)
    inp = dict(h=h, root=root, We=We, be=be)
    report(f"gcn aggregate D{D}", hip, run(f, inp, w, torch.float32), run(f, inp, w, torch.float64))


def all_cases():
    linear_case(31598, 300, 300)
    linear_case(31598, 128, 600)
    linear_case(256, 25012, 128)
    ln_case(31855, 128)
    bn_case(31598, 300, True)
    bn_case(256, 600, True)
    attn_case(list(np.random.default_rng(0).integers(20, 400, 64)), 128, 4)
    attn_case([513] * 8, 256, 4)
    agg_case(256, 300)
    return ROWS


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    all_cases()
