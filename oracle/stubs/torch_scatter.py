"""Container-only stub of torch-scatter 2.0.6 `scatter` (requirement.yml:98).

Published semantics restated: out[index[i]] (reduce)= src[i] along `dim`, output size
`dim_size`; empty segments yield 0 for every reduce (sum/mean/min/max).
Test infrastructure only (see oracle/stubs/README.md).
"""
import torch


def _expand_index(index, src, dim):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        index = index.view(shape)
    return index.expand_as(src), dim


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    index, dim = _expand_index(index, src, dim)
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    if reduce in ("sum", "add"):
        return torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, index, src)
    if reduce == "mean":
        s = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, index, src)
        c = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, index, torch.ones_like(src))
        return s / c.clamp(min=1)
    if reduce in ("min", "max"):
        o = torch.zeros(shape, dtype=src.dtype, device=src.device)
        o = o.scatter_reduce(dim, index, src, reduce="amin" if reduce == "min" else "amax", include_self=False)
        return o
    raise ValueError(reduce)
