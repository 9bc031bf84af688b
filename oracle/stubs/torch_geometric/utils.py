"""Stub of torch_geometric.utils.degree (PyG 1.6.3): count of occurrences of each index."""
import torch


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros((n,), dtype=dtype, device=index.device)
    return out.scatter_add_(0, index, out.new_ones((index.size(0),)))
