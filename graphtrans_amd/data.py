"""Minimal stand-in for the PyG `Batch` object the reference's modules consume.

The reference reads only attributes (`x, edge_index, edge_attr, batch, node_depth, y, y_arr,
adj_list`; modules/gnn_module.py:61-62,173-174, models/gnn_transformer.py:95,103), so a plain
attribute bag with `.to(device)` is a drop-in for the hot path.
"""
import ctypes

import torch


def _stream():   # (graph.py imports this module's Batch lazily; keep data.py free of package imports at module level)
    from .graph import _stream as f
    return f()


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


class _StoreDesc(ctypes.Structure):  # gt_graph_store (include/graphtrans_hip.h)
    _fields_ = [(k, ctypes.c_void_p) for k in ("node_ptr", "edge_ptr", "x", "node_depth", "edge_src", "edge_dst",
                                                "edge_attr", "attr_rank", "y")] + \
               [("y_row_bytes", ctypes.c_int64), ("num_graphs", ctypes.c_int64), ("x_cols", ctypes.c_int32),
                ("ea_cols", ctypes.c_int32)]


class _CollateOut(ctypes.Structure):  # gt_collate_out
    _fields_ = [(k, ctypes.c_void_p) for k in ("x", "node_depth", "batch", "ptr", "edge_index", "edge_attr_f32",
                                                "edge_attr_i64", "y")]


class GraphStore:
    """A whole dataset resident in HBM, mini-batches assembled on the device (`gt_collate`).

    Stands where the reference has `dataset[idx]` + per-sample `augment_edge` (dataset/utils.py:89-141,
    dataset/code.py:97-101) + PyG's DataLoader collation (main.py:149-152): `collate(ids)` returns the
    same `Batch` attributes (x, edge_index, edge_attr, batch, node_depth, y | y_arr) for the graphs `ids`
    in that order.  If the graphs carry `node_is_attributed` the Code2 augmentation is applied
    (edges [ast, ast^-1, next-token, next-token^-1], float (E,2) edge_attr); otherwise edges and int64
    edge_attr are copied as stored.  Per-graph sizes stay on the host, so sizing a batch costs no sync.
    """

    def __init__(self, graphs, device="cuda", label_key=None):
        import numpy as np
        from . import _lib
        if not torch.cuda.is_available():
            raise RuntimeError("graphtrans_amd.GraphStore lives in GPU memory (no CPU fallback)")
        self._np, self._lib = np, _lib
        dev = torch.device(device)
        g0 = graphs[0]
        self.label_key = label_key or ("y_arr" if "y_arr" in g0 else ("y" if "y" in g0 else None))
        self.nodes = np.array([g["x"].shape[0] for g in graphs], np.int64)
        self.edges = np.array([g["edge_index"].shape[1] for g in graphs], np.int64)
        self.augment = "node_is_attributed" in g0
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        cat = lambda k, ax=0: np.concatenate([np.asarray(g[k]) for g in graphs], ax)  # noqa: E731
        self.node_ptr = up(np.concatenate([[0], np.cumsum(self.nodes)]).astype(np.int64))
        self.edge_ptr = up(np.concatenate([[0], np.cumsum(self.edges)]).astype(np.int64))
        self.x = up(cat("x").astype(np.int64))
        ei = cat("edge_index", 1).astype(np.int64)
        self.edge_src, self.edge_dst = up(ei[0]), up(ei[1])
        self.node_depth = up(cat("node_depth").reshape(-1).astype(np.int64)) if "node_depth" in g0 else None
        self.edge_attr = up(cat("edge_attr").astype(np.int64)) if ("edge_attr" in g0 and not self.augment) else None
        self.y = up(cat(self.label_key)) if self.label_key else None
        self.attr_rank = None
        self.out_edges = self.edges
        if self.augment:
            flag = cat("node_is_attributed").reshape(-1).astype(np.int64)
            cnt = np.add.reduceat((flag == 1).astype(np.int64), np.cumsum(self.nodes) - self.nodes) if len(graphs) else flag[:0]
            self.out_edges = 2 * self.edges + 2 * np.maximum(cnt - 1, 0)
            dflag = up(flag)
            self.attr_rank = torch.empty(flag.size + 1, dtype=torch.int64, device=dev)
            _lib.launch("gt_attr_rank", dflag.data_ptr(), flag.size, self.attr_rank.data_ptr(),
                        _stream())
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        self._desc = _StoreDesc(p(self.node_ptr), p(self.edge_ptr), p(self.x), p(self.node_depth), p(self.edge_src),
                                p(self.edge_dst), p(self.edge_attr), p(self.attr_rank), p(self.y),
                                0 if self.y is None else self.y[0].numel() * self.y.element_size(), len(graphs),
                                self.x.shape[1], 0 if self.edge_attr is None else self.edge_attr.shape[1])
        self.device = dev

    def __len__(self):
        return int(self.nodes.size)

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in (self.node_ptr, self.edge_ptr, self.x, self.edge_src, self.edge_dst,
                                                            self.node_depth, self.edge_attr, self.y, self.attr_rank) if t is not None)

    def collate(self, ids):
        """ids: host sequence / numpy / CPU tensor of graph indices (a sampler's output)."""
        np = self._np
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        if ids.size and (ids.min() < 0 or ids.max() >= len(self)):
            raise IndexError("graph id out of range")
        B = int(ids.size)
        sizes = self.nodes[ids]
        N, E = int(sizes.sum()), int(self.out_edges[ids].sum())
        dev = self.device
        i64 = dict(dtype=torch.int64, device=dev)
        d_ids = torch.from_numpy(ids).to(dev, non_blocking=True)
        x = torch.empty(N, self.x.shape[1], **i64)
        batch = torch.empty(N, **i64)
        edge_index = torch.empty(2, E, **i64)
        depth = torch.empty(N, 1, **i64) if self.node_depth is not None else None
        ea_f = torch.empty(E, 2, dtype=torch.float32, device=dev) if self.augment else None
        ea_i = torch.empty(E, self.edge_attr.shape[1], **i64) if (self.edge_attr is not None and not self.augment) else None
        y = torch.empty((B,) + tuple(self.y.shape[1:]), dtype=self.y.dtype, device=dev) if self.y is not None else None
        L = self._lib.lib()
        ws_bytes = L.gt_collate_workspace_bytes(B)
        ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        out = _CollateOut(p(x), p(depth), p(batch), None, p(edge_index), p(ea_f), p(ea_i), p(y))
        self._lib.launch("gt_collate", ctypes.byref(self._desc), p(d_ids), B, N, E, ctypes.byref(out), p(ws), ws_bytes,
                         _stream())
        b = Batch(x=x, edge_index=edge_index, batch=batch, edge_attr=ea_f if self.augment else ea_i)
        if depth is not None:
            b.node_depth = depth
        if y is not None:
            setattr(b, self.label_key, y)
        b._num_graphs, b._sizes = B, sizes
        return b
