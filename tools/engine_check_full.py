#!/usr/bin/env python
"""Race check at the benchmark's size: the fused path's gradients on the full Code2 b256 / Molpcba b256 configuration
(dropout off) must be bitwise identical run after run and match the module path; repeated to shake out ordering bugs
between the main, virtual-node and dW streams.
GT_CHECK_ITERS (20), GT_CHECK_MODE (bf16 | mixed), GT_CHECK_WORKLOADS (code2,molpcba), GT_CHECK_NOSYNC=1: no device synchronisation
between the passes (the host runs ahead as in training: consecutive passes overlap on the side streams), GT_CHECK_DROPOUT=1: dropout
stays at the configuration's rates and torch's generator (the source of the per-step dropout seeds) is re-seeded before every pass."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from graphtrans_amd import ops

dev = torch.device("cuda:0")
ITERS = int(os.environ.get("GT_CHECK_ITERS", "20"))
GNN = torch.float32 if os.environ.get("GT_CHECK_MODE", "bf16") == "mixed" else torch.bfloat16   # GNN-side GEMM arithmetic
NOSYNC = os.environ.get("GT_CHECK_NOSYNC") == "1"
DROPOUT = os.environ.get("GT_CHECK_DROPOUT") == "1"
for wl in os.environ.get("GT_CHECK_WORKLOADS", "code2,molpcba").split(","):
    ops.set_matmul_dtype(GNN)
    torch.manual_seed(0)
    args, model, gen, loss_fn, _ = bench.build(wl, torch.bfloat16, dev, 256)
    if not DROPOUT:
        for m in model.modules():
            if hasattr(m, "dropout_p"):
                m.dropout_p = 0.0
        model.gnn_node.drop_ratio = 0.0
    model.train()
    b = bench.attach_sizes(gen(0)).to(dev)

    def grads(m, fused):
        m.fused = fused
        for p in m.parameters():
            p.grad = None
        b.__dict__.pop("_gt_structure", None)
        if DROPOUT:
            torch.manual_seed(1234)
        loss_fn(m(b), b).backward()
        if not NOSYNC:
            torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in m.parameters()]

    ref_model = copy.deepcopy(model)
    ref = grads(ref_model, False)
    first = None
    worst = 0.0
    for it in range(ITERS):
        g = grads(model, True)
        if first is None:
            first = g
            for a, r, (n, _) in zip(g, ref, model.named_parameters()):
                e = (a - r).abs().max().item() / max(1e-6, r.abs().max().item())
                worst = max(worst, e)
        else:
            bad = [(n, (a - f).abs().max().item()) for a, f, (n, _) in zip(g, first, model.named_parameters()) if not torch.equal(a, f)]
            if bad:
                print(wl, "NOT REPRODUCIBLE at iteration", it, len(bad), "of", len(g), "parameters differ:", bad[:6])
                if os.environ.get("GT_CHECK_VERBOSE"):
                    badn = {n for n, _ in bad}
                    for n, _ in model.named_parameters():
                        print("   ", "DIFF" if n in badn else "same", n)
                sys.exit(1)
    print(wl, ITERS, "fused backward passes bitwise identical; max rel. difference to the module path %.2e" % worst)
