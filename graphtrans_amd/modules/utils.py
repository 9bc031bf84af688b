"""pad_batch / unpad_batch with the reference's signatures (modules/utils.py:5-53), as single
gather/scatter kernels over graph_ptr instead of a Python loop over graphs with O(B) syncs."""
import torch

from .. import ops
from ..graph import GraphStructure


def _structure_from_batch(batch):
    ei = torch.zeros((2, 0), dtype=torch.int64, device=batch.device)
    return GraphStructure.build(ei, batch)


def pad_batch(h_node, batch, max_input_len, get_mask=False, graph=None):
    """-> padded (S,B,d), src_padding_mask (B,S) bool (True = padding)
       [, num_nodes (B,), layout, max_num_nodes] when get_mask.
    Differences from the reference return: `masks` (the per-graph boolean node masks, only ever
    handed back to unpad_batch) is replaced by the SeqLayout object unpad_batch needs."""
    gs = graph if graph is not None else _structure_from_batch(batch)
    lay = gs.layout("padded", int(max_input_len), False)
    tokens, mask = ops.seq_gather(h_node, None, gs, lay, want_mask=True)
    padded = tokens.view(lay.S, gs.B, h_node.shape[-1])
    if get_mask:
        num_nodes = (gs.graph_ptr[1:] - gs.graph_ptr[:-1]).to(torch.int64)
        return padded, mask, num_nodes, (gs, lay), lay.S
    return padded, mask


def unpad_batch(padded_h_node, prev_h_node, num_nodes, origin_mask, max_num_nodes):
    """Inverse of pad_batch; `origin_mask` is the (structure, layout) pair pad_batch returned.
    Nodes truncated by pad_batch keep their prev_h_node rows (modules/utils.py:41-52)."""
    gs, lay = origin_mask
    tokens = padded_h_node.reshape(-1, padded_h_node.shape[-1])
    return ops.seq_scatter(tokens, prev_h_node, gs, lay)
