#!/usr/bin/env python
"""gt_aggregate_fwd / _bwd alone on the Molpcba- and Code2-shaped batches (C-side launch profiler, 20 calls)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from graphtrans_amd import _lib, ops, synth
from graphtrans_amd.encoders import BondEncoder
from graphtrans_amd.graph import GraphStructure
from graphtrans_amd.modules.conv import edge_spec, _materialize

DEV = "cuda:0"
D = 300


def run(name, b, conv, enc):
    b = b.to(DEV)
    gs = GraphStructure.build(b.edge_index, b.batch, num_graphs=b.num_graphs)
    h = torch.randn(gs.N, D, device=DEV, requires_grad=True)
    sp = torch.randn(1, D, device=DEV, requires_grad=True) if conv == "gcn" else torch.zeros(1, device=DEV, requires_grad=True)
    spec = _materialize(edge_spec(enc, b.edge_attr, D))
    g = torch.randn(gs.N, D, device=DEV)
    for it in range(25):
        if it == 5:
            _lib.profile_enable(1)
        out = ops.aggregate(h, gs, conv, sp, spec)
        out.backward(g)
        spec.tables = None if spec.kind != "tables" else torch.cat(spec.table_list, dim=0)
    rec = _lib.profile_records()
    _lib.profile_enable(0)
    agg = {}
    for n, ms, dims in rec:
        a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += ms
    print(name, f"N={gs.N} E={gs.E}", {k: round(1e3 * v[1] / v[0], 1) for k, v in agg.items()}, "us")


torch.manual_seed(0)
mol = synth.molpcba_like(B=256, seed=0)
run("molpcba GIN tables", mol, "gin", BondEncoder(D).to(DEV))
lin = torch.nn.Linear(3, D).to(DEV)
mol2 = synth.molpcba_like(B=256, seed=0); mol2.edge_attr = mol2.edge_attr.float()
run("molpcba GIN linear(3)", mol2, "gin", lin)
run("molpcba GIN no edge enc", mol, "gin", lambda d: 0)
c2 = synth.code2_like(B=256, seed=0)
run("code2 GCN linear(2)", c2, "gcn", torch.nn.Linear(2, D).to(DEV))
c2s = synth.code2_like(B=54, seed=0)
run("code2 (54 graphs ~ molpcba N) GCN linear(2)", c2s, "gcn", torch.nn.Linear(2, D).to(DEV))
