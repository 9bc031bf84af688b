"""The trainers' calc_loss functions (define the backward seed of the fwd+bwd metric)."""
import torch


def code2_loss(pred_list, y_arr):
    """dataset/code.py:39-45: mean over the max_seq_len heads of CrossEntropy(pred_i, y_arr[:, i]) on gt_xent_* (the heads of
    the HIP path come out of one GEMM as one (B, L, C) tensor; separate (B, C) heads go through the same kernels one by one)."""
    from . import ops
    stacked = getattr(pred_list, "stacked", None)
    if stacked is not None:  # heads computed as one GEMM: equal-size means -> one cross-entropy over B*L rows
        B, L, C = stacked.shape
        if not stacked.is_cuda:
            raise RuntimeError("graphtrans_amd.losses.code2_loss runs on the GPU only (no CPU fallback); "
                               "the CPU statement is oracle/reference_math.py:code2_loss")
        if stacked.dtype != torch.float32 or stacked.stride(2) != 1 or stacked.stride(1) != C:
            stacked = stacked.to(torch.float32).contiguous()
        return ops.softmax_xent(stacked, y_arr)
    loss = 0
    for i, pred in enumerate(pred_list):
        if not pred.is_cuda:
            raise RuntimeError("graphtrans_amd.losses.code2_loss runs on the GPU only (no CPU fallback); "
                               "the CPU statement is oracle/reference_math.py:code2_loss")
        p32 = pred.to(torch.float32)
        if p32.stride(-1) != 1:
            p32 = p32.contiguous()
        loss = loss + ops.softmax_xent(p32.unsqueeze(1), y_arr[:, i:i + 1])
    return loss / len(pred_list)


def dp_label_denominator(y, group=None):
    """Global number of labelled entries / world size as a 1-element fp32 tensor (None when not data parallel)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    den = (y == y).sum().to(torch.float32).reshape(1)
    dist.all_reduce(den, group=group)
    return den / dist.get_world_size(group)


def mol_loss(pred, y, group=None):
    """dataset/mol.py:24-31: BCE-with-logits over labelled (non-NaN) entries only, mean over them
    (`ops.masked_bce`: no boolean indexing, which would be a device->host sync on the number of kept entries).

    Data parallel (torch.distributed initialised, world > 1): the denominator is the GLOBAL labelled count / world,
    so that the all-reduce AVERAGE of the rank gradients is the gradient of the global-batch loss;
    the returned value is this rank's share, their mean over ranks is the global loss."""
    if pred.is_cuda and y.is_cuda:
        from . import ops
        return ops.masked_bce(pred.to(torch.float32), y.to(torch.float32), dp_label_denominator(y, group))
    raise RuntimeError("graphtrans_amd.losses.mol_loss runs on the GPU only (no CPU fallback); "
                       "the CPU statement is oracle/reference_math.py:mol_loss")


def tud_loss(pred, y):
    """dataset/tud.py:25-27: CrossEntropyLoss()(pred, y) = the one-head case of the fused cross-entropy."""
    if pred.is_cuda and pred.dim() == 2 and y.dim() == 1:
        from . import ops
        p = pred.to(torch.float32)
        if p.stride(1) != 1:
            p = p.contiguous()
        return ops.softmax_xent(p.unsqueeze(1), y.reshape(-1, 1))
    raise RuntimeError("graphtrans_amd.losses.tud_loss runs on the GPU only (no CPU fallback); "
                       "the CPU statement is oracle/reference_math.py:tud_loss")
