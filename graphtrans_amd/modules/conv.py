"""GINConv / GCNConv with the reference's parameters (modules/conv.py:10-71), message passing by
the fused HIP aggregate kernel (graphtrans_amd/csrc/aggregate.hip) instead of PyG MessagePassing.
"""
import torch

from .. import ops
from ..graph import GraphStructure
from .norm import BatchNorm1d, mlp_bn_relu

_LDS_TABLE_BUDGET = 160 * 1024  # LDS per CU


def tables_fit_lds(rows, emb_dim):
    """The backward kernel keeps 4 per-wave copies of the table-gradient rows plus the tables themselves (sharing
    the space of the 4-row reduction stage) in LDS: csrc/aggregate.hip:bwd_lds_bytes."""
    return (4 * rows + max(4, rows)) * emb_dim * 4 <= _LDS_TABLE_BUDGET


def edge_spec(edge_encoder, edge_attr, emb_dim):
    """Map an arbitrary reference-style edge encoder onto a kernel edge mode (see gt_edge_mode)."""
    if not isinstance(edge_encoder, torch.nn.Module):
        e = edge_encoder(edge_attr)  # dataset/tud.py:67-71 returns python 0
        if isinstance(e, (int, float)) and e == 0:
            return ops.EdgeSpec("none")
        return ops.EdgeSpec("dense", dense=e)
    if isinstance(edge_encoder, torch.nn.Linear) and edge_encoder.in_features <= 4 and edge_encoder.bias is not None:
        return ops.EdgeSpec("linear", attr=edge_attr, weight=edge_encoder.weight, bias=edge_encoder.bias)
    tabs = getattr(edge_encoder, "bond_embedding_list", None)
    if tabs is not None and edge_attr.dtype == torch.int64 and edge_attr.shape[1] <= 4:
        k = edge_attr.shape[1]
        rows = sum(int(t.weight.shape[0]) for t in list(tabs)[:k])
        if tables_fit_lds(rows, emb_dim):
            off, acc = [], 0
            for t in list(tabs)[:k]:
                off.append(acc)
                acc += int(t.weight.shape[0])
            spec = ops.EdgeSpec("tables", attr=edge_attr, tables=None, tab_off=off)
            spec.table_list = [t.weight for t in list(tabs)[:k]]
            return spec
    return ops.EdgeSpec("dense", dense=edge_encoder(edge_attr))


def _materialize(spec):
    """Concatenate the embedding tables for the fine-grained kernel call (autograd splits the grad)."""
    if spec.kind == "tables" and spec.tables is None:
        spec.tables = torch.cat(spec.table_list, dim=0)
    return spec


def _structure(x, edge_index, graph):
    if graph is not None:
        return graph
    batch = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device)
    return GraphStructure.build(edge_index, batch, num_graphs=1)


class GINConv(torch.nn.Module):
    def __init__(self, emb_dim: int, edge_encoder_cls):
        super().__init__()
        self.mlp = torch.nn.Sequential(
            torch.nn.Linear(emb_dim, 2 * emb_dim), BatchNorm1d(2 * emb_dim), torch.nn.ReLU(),
            torch.nn.Linear(2 * emb_dim, emb_dim))
        self.eps = torch.nn.Parameter(torch.Tensor([0]))
        self.edge_encoder = edge_encoder_cls(emb_dim)
        self.emb_dim = emb_dim

    def forward(self, x, edge_index, edge_attr, graph=None):
        gs = _structure(x, edge_index, graph)
        spec = edge_spec(self.edge_encoder, edge_attr, self.emb_dim)
        # (1 + eps) * x + sum_k relu(x_j + e_k), fused  (conv.py:28,33)
        return mlp_bn_relu(self.mlp, ops.aggregate(x, gs, "gin", self.eps, _materialize(spec)))


class GCNConv(torch.nn.Module):
    def __init__(self, emb_dim, edge_encoder_cls):
        super().__init__()
        self.linear = torch.nn.Linear(emb_dim, emb_dim)
        self.root_emb = torch.nn.Embedding(1, emb_dim)
        self.edge_encoder = edge_encoder_cls(emb_dim)
        self.emb_dim = emb_dim

    def forward(self, x, edge_index, edge_attr, graph=None):
        gs = _structure(x, edge_index, graph)
        x = ops.linear_module(self.linear, x)
        spec = edge_spec(self.edge_encoder, edge_attr, self.emb_dim)
        # sum_k norm_k relu(x_j + e_k) + relu(x + root_emb) / deg, fused  (conv.py:54-68)
        return ops.aggregate(x, gs, "gcn", self.root_emb.weight, _materialize(spec))
