#!/usr/bin/env python
"""Per-parameter gradient difference between the fused model path (engine.py) and the module path."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_engine import _args, _run, DEV
from graphtrans_amd import synth
from graphtrans_amd.encoders import ASTNodeEncoder
from graphtrans_amd.models.gnn_transformer import GNNTransformer
kw = eval(sys.argv[1]) if len(sys.argv) > 1 else {}
args = _args(**kw)
torch.manual_seed(0)
model = GNNTransformer(50, ASTNodeEncoder(64, 98, 300, 20), lambda d: torch.nn.Linear(2, d), args).to(DEV)
with torch.no_grad():
    for p in model.parameters():
        if p.dim() == 1: p.add_(torch.randn_like(p) * 0.1)
    if args.gnn_virtual_node: model.gnn_node.virtualnode_embedding.weight.normal_(0, 0.3)
model.train()
b = synth.code2_like(B=12, seed=5, num_nodeattributes=300).to(DEV)
y = torch.randint(0, 50, (12, 5), device=DEV)
ref = copy.deepcopy(model)
l0, g0, b0 = _run(ref, b, y, False, 7)
l1, g1, b1 = _run(model, b, y, True, 7)
print("loss", l0.item(), l1.item())
for n in g0:
    e = (g0[n] - g1[n]).abs().max().item(); s = g0[n].abs().max().item()
    bad = e > 1e-4 * max(s, 1e-6) + 1e-7
    if bad or "-v" in sys.argv: print(f"{n:70s} ref_max {s:.3e} err {e:.3e} {'BAD' if bad else ''}")
