"""The trainers' calc_loss functions (define the backward seed of the fwd+bwd metric)."""
import torch
import torch.nn.functional as F


def code2_loss(pred_list, y_arr):
    """dataset/code.py:39-45: mean over the max_seq_len heads of CrossEntropy(pred_i, y_arr[:, i])."""
    stacked = getattr(pred_list, "stacked", None)
    if stacked is not None:  # heads computed as one GEMM: equal-size means -> one cross-entropy over B*L rows
        B, L, C = stacked.shape
        if stacked.is_cuda and stacked.dtype == torch.float32 and stacked.stride(2) == 1 and stacked.stride(1) == C:
            from . import ops
            return ops.softmax_xent(stacked, y_arr)
        return F.cross_entropy(stacked.to(torch.float32).reshape(B * L, C), y_arr[:, :L].reshape(B * L))
    loss = 0
    for i, pred in enumerate(pred_list):
        loss = loss + F.cross_entropy(pred.to(torch.float32), y_arr[:, i])
    return loss / len(pred_list)


def mol_loss(pred, y):
    """dataset/mol.py:24-31: BCE-with-logits over labelled (non-NaN) entries only.  Written without
    boolean indexing (`pred[is_labeled]` is a device->host sync on the number of kept entries): the
    mean over the labelled entries is sum(mask * bce) / sum(mask)."""
    pred = pred.to(torch.float32)
    is_labeled = y == y
    target = torch.where(is_labeled, y.to(torch.float32), torch.zeros((), dtype=torch.float32, device=y.device))
    per = F.binary_cross_entropy_with_logits(pred, target, reduction="none")
    m = is_labeled.to(torch.float32)
    return (per * m).sum() / m.sum()


def tud_loss(pred, y):
    """dataset/tud.py:25-27."""
    return F.cross_entropy(pred, y)
