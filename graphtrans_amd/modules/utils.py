"""pad_batch / unpad_batch with the reference's signatures (modules/utils.py:5-53), as single
gather/scatter kernels over graph_ptr instead of a Python loop over graphs with O(B) syncs."""
from collections.abc import Sequence

import torch

from .. import ops
from ..graph import GraphStructure


class NodeMasks(Sequence):
    """The `masks` value of pad_batch(get_mask=True) (modules/utils.py:8-13,28): a sequence of B boolean node masks,
    masks[i] == batch.eq(i).  The masks are materialised on access (one compare each, as the reference builds them);
    unpad_batch does not read them -- it takes the structure / layout pad_batch already built from this object."""

    def __init__(self, batch, structure, layout):
        self.batch, self.structure, self.layout = batch, structure, layout

    def __len__(self):
        return int(self.structure.B)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < -n or i >= n:
            raise IndexError("mask index out of range")
        return self.batch.eq(i % n)


def _structure_from_batch(batch):
    ei = torch.zeros((2, 0), dtype=torch.int64, device=batch.device)
    return GraphStructure.build(ei, batch)


def pad_batch(h_node, batch, max_input_len, get_mask=False, graph=None):
    """-> padded (S,B,d), src_padding_mask (B,S) bool (True = padding)
       [, num_nodes, masks, max_num_nodes] when get_mask -- the reference's five values (modules/utils.py:27-29):
    num_nodes[i] is graph i's node count (a (B,) int64 tensor instead of a list of 0-d tensors), masks[i] is
    batch.eq(i) (NodeMasks: built on access), max_num_nodes = S = min(max nodes, max_input_len)."""
    gs = graph if graph is not None else _structure_from_batch(batch)
    lay = gs.layout("padded", int(max_input_len), False)
    tokens, mask = ops.seq_gather(h_node, None, gs, lay, want_mask=True)
    padded = tokens.view(lay.S, gs.B, h_node.shape[-1])
    if get_mask:
        num_nodes = (gs.graph_ptr[1:] - gs.graph_ptr[:-1]).to(torch.int64)
        return padded, mask, num_nodes, NodeMasks(batch, gs, lay), lay.S
    return padded, mask


def unpad_batch(padded_h_node, prev_h_node, num_nodes, origin_mask, max_num_nodes):
    """Inverse of pad_batch (modules/utils.py:32-53).  `origin_mask` is pad_batch's `masks` (NodeMasks: the structure
    and layout ride along) or any sequence of B boolean node masks (the graph vector is rebuilt from them).
    Nodes truncated by pad_batch keep their prev_h_node rows (modules/utils.py:41-52)."""
    if isinstance(origin_mask, NodeMasks):
        gs, lay = origin_mask.structure, origin_mask.layout
        if int(max_num_nodes) != lay.S:
            lay = gs.layout("padded", int(max_num_nodes), False)
    elif isinstance(origin_mask, tuple) and len(origin_mask) == 2 and isinstance(origin_mask[0], GraphStructure):
        gs, lay = origin_mask
    else:
        m = torch.stack([x.reshape(-1) for x in origin_mask]).to(torch.uint8)   # (B, N): one graph per node
        gs = _structure_from_batch(m.argmax(0))
        lay = gs.layout("padded", int(max_num_nodes), False)
    tokens = padded_h_node.reshape(-1, padded_h_node.shape[-1])
    return ops.seq_scatter(tokens, prev_h_node, gs, lay)
