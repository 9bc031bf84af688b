"""BaseModel contract (models/base_model.py:5-25)."""
import torch.nn as nn


class BaseModel(nn.Module):
    @staticmethod
    def need_deg():
        return False

    @staticmethod
    def add_args(parser):
        return

    @staticmethod
    def name(args):
        raise NotImplementedError

    def __init__(self):
        super().__init__()

    def forward(self, batched_data, perturb=None):
        raise NotImplementedError

    def epoch_callback(self, epoch):
        return


class StackedHeads(list):
    """list of the per-position logits (B, num_tasks), all views of `.stacked` (B, L, num_tasks)."""

    def __init__(self, stacked):
        super().__init__(stacked[:, i] for i in range(stacked.shape[1]))
        self.stacked = stacked


def stacked_heads(h_graph, heads, num_tasks):
    """The max_seq_len prediction heads (models/gnn_transformer.py:124-126,
    models/pna_transformer.py:94-96) as ONE GEMM over the stacked weights; the returned list holds views
    of it (losses.code2_loss recognises them and runs one fused cross-entropy).  5 x 5002 logits per
    graph are not a multiple of 4, so the rows of the logits buffer are padded (gt_linear_fwd_ld)."""
    import torch

    from .. import ops
    w = torch.cat([m.weight for m in heads], dim=0)
    b = torch.cat([m.bias for m in heads], dim=0)
    N = w.shape[0]
    h_graph = h_graph.float()
    if not (h_graph.is_cuda and h_graph.shape[-1] % 4 == 0):
        raise RuntimeError("stacked_heads: GPU tensor with a feature size that is a multiple of 4 expected")
    stacked = ops.linear(h_graph, w, b, ldy=(N + 3) // 4 * 4).view(h_graph.shape[0], len(heads), num_tasks)
    return StackedHeads(stacked)
