"""Minimal stand-in for the PyG `Batch` object the reference's modules consume.

The reference reads only attributes (`x, edge_index, edge_attr, batch, node_depth, y, y_arr,
adj_list`; modules/gnn_module.py:61-62,173-174, models/gnn_transformer.py:95,103), so a plain
attribute bag with `.to(device)` is a drop-in for the hot path.
"""
import torch


class Batch:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device, non_blocking=False):
        out = Batch()
        for k, v in self.__dict__.items():
            out.__dict__[k] = v.to(device, non_blocking=non_blocking) if isinstance(v, torch.Tensor) else v
        return out

    @property
    def num_graphs(self):
        if "_num_graphs" in self.__dict__:
            return self._num_graphs
        return int(self.batch[-1]) + 1

    @property
    def num_nodes(self):
        return self.batch.numel()

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]
