"""Stub of the parts of torch_geometric.nn (1.6.3) the reference imports.

`MessagePassing` restates PyG's documented default flow ("source_to_target"):
  * message() arguments named `<k>_j` are gathered as kwargs[k][edge_index[0]] (source),
    `<k>_i` as kwargs[k][edge_index[1]] (target); other names are passed through;
  * aggregate() scatter-reduces the messages at edge_index[1] with dim_size = N;
  * update() receives the aggregate.
Test infrastructure only (see oracle/stubs/README.md).
"""
import inspect

import torch
from torch_scatter import scatter


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=0):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr = aggr
        self.node_dim = node_dim
        self._msg_params = [p for p in inspect.signature(self.message).parameters]
        self._aggr_params = [p for p in inspect.signature(self.aggregate).parameters]

    def propagate(self, edge_index, size=None, **kwargs):
        n = None
        for name in self._msg_params:
            if name.endswith("_j") or name.endswith("_i"):
                n = kwargs[name[:-2]].size(self.node_dim)
                break
        msg_kwargs = {}
        for name in self._msg_params:
            if name.endswith("_j"):
                msg_kwargs[name] = kwargs[name[:-2]].index_select(self.node_dim, edge_index[0])
            elif name.endswith("_i"):
                msg_kwargs[name] = kwargs[name[:-2]].index_select(self.node_dim, edge_index[1])
            else:
                msg_kwargs[name] = kwargs.get(name)
        out = self.message(**msg_kwargs)
        aggr_kwargs = {}
        if "index" in self._aggr_params:
            aggr_kwargs["index"] = edge_index[1]
        if "dim_size" in self._aggr_params:
            aggr_kwargs["dim_size"] = n
        out = self.aggregate(out, **aggr_kwargs)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, dim_size=None):
        reduce = "sum" if self.aggr == "add" else self.aggr
        return scatter(inputs, index, dim=self.node_dim, dim_size=dim_size, reduce=reduce)

    def update(self, inputs):
        return inputs


def global_add_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    return scatter(x, batch, dim=0, dim_size=size, reduce="sum")


def global_mean_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    return scatter(x, batch, dim=0, dim_size=size, reduce="mean")


def global_max_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    return scatter(x, batch, dim=0, dim_size=size, reduce="max")


class _Unavailable(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("not needed on the GraphTrans hot path; stubbed for import only")


class GlobalAttention(_Unavailable):
    pass


class Set2Set(_Unavailable):
    pass


class BatchNorm(torch.nn.Module):
    """PyG 1.6.3 `torch_geometric.nn.norm.BatchNorm`: an nn.BatchNorm1d held as `.module` (so the
    state_dict keys read `...batch_norms.<i>.module.weight`), forward = self.module(x)."""

    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.module = torch.nn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)

    def reset_parameters(self):
        self.module.reset_parameters()

    def forward(self, x):
        return self.module(x)


class PNAConv(_Unavailable):
    """PyG's PNAConv is third-party code absent from /root/reference; not restated here.
    oracle/make_golden.py:g12_pna rebinds this name to the reference's own in-tree statement of the
    class (modules/pna_layer.py:20-171) before importing modules/pna/pna_module.py."""
