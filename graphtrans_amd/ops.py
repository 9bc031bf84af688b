"""torch.autograd.Function wrappers over the C ABI (include/graphtrans_hip.h).

Every function here launches hand-written gfx950 kernels through ctypes on torch's current
stream; tensors are only containers for device memory.  No eager/CPU fallback exists: a CPU
tensor or a missing library raises.
"""
import ctypes as C

import os

import torch

from . import _lib
from ._lib import GT_BF16, GT_CONV_GCN, GT_CONV_GIN, GT_EDGE_DENSE, GT_EDGE_LINEAR, GT_EDGE_NONE, GT_EDGE_TABLES, GT_F32
from .graph import _ptr, _stream


def _dtype_code(t):
    if t.dtype == torch.float32:
        return GT_F32
    if t.dtype == torch.bfloat16:
        return GT_BF16
    raise TypeError(f"graphtrans_amd kernels take float32 or bfloat16, got {t.dtype}")


def _dev(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: graphtrans_amd has no CPU fallback")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _f32(t):
    return None if t is None else _dev(t.detach().float(), "param")


# ------------------------------------------------------------------------------------------------
# fused message passing
# ------------------------------------------------------------------------------------------------
class EdgeSpec:
    """How the kernel obtains e_k.  kind in {none, linear, tables, dense} (gt_edge_mode)."""

    def __init__(self, kind, attr=None, weight=None, bias=None, tables=None, tab_off=None, dense=None):
        self.kind, self.attr, self.weight, self.bias = kind, attr, weight, bias
        self.tables, self.tab_off, self.dense = tables, tab_off, dense
        self.table_list = None


class _Aggregate(torch.autograd.Function):
    """out = conv-aggregate(h) (gt_aggregate_fwd / gt_aggregate_bwd).
    Differentiable inputs: h, self_param (root_emb.weight | eps), and the edge-encoder tensors
    (Linear weight+bias | concatenated tables | dense edge embedding)."""

    @staticmethod
    def forward(ctx, h, self_param, ew, eb, dense, gs, conv, mode, attr, tab_off):
        h = _dev(h, "h")
        L = _lib.lib()
        N, D = h.shape
        out = torch.empty_like(h)
        sp = _f32(self_param)
        ew32, eb32 = _f32(ew), _f32(eb)
        dense_c = None if dense is None else _dev(dense.to(h.dtype), "edge embedding")
        attr_c = None if attr is None else _dev(attr, "edge_attr")
        K = 0 if attr_c is None else int(attr_c.shape[1])
        toff = (C.c_int32 * max(len(tab_off), 1))(*tab_off) if tab_off else None
        meta = dict(N=N, E=gs.E, D=D, elt=h.element_size(),
                    attr_bytes=0 if attr_c is None else attr_c.shape[1] * attr_c.element_size())
        _lib.launch("gt_aggregate_fwd", conv, mode, _dtype_code(h), _ptr(h), N, gs.E, D, _ptr(gs.in_ptr),
                    _ptr(gs.in_src), _ptr(gs.in_eid), _ptr(gs.deg), _ptr(gs.dis), _ptr(sp), _ptr(attr_c), K,
                    _ptr(ew32), _ptr(eb32), toff, (int(ew32.shape[0]) if (tab_off and ew32 is not None) else 0), _ptr(dense_c),
                    _ptr(out), _stream(), meta=meta)
        ctx.meta = meta
        ctx.save_for_backward(h, sp, ew32, eb32, dense_c, attr_c)
        ctx.gs, ctx.conv, ctx.mode, ctx.tab_off, ctx.K = gs, conv, mode, tab_off, K
        ctx.param_dtypes = (self_param.dtype, None if ew is None else ew.dtype, None if eb is None else eb.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        h, sp, ew32, eb32, dense_c, attr_c = ctx.saved_tensors
        gs, conv, mode, K = ctx.gs, ctx.conv, ctx.mode, ctx.K
        L = _lib.lib()
        g = _dev(g.to(h.dtype), "grad_out")
        N, D = h.shape
        dh = torch.empty_like(h)
        dev = h.device
        rows = 0 if mode != GT_EDGE_TABLES else int(ew32.shape[0])
        d_self = torch.empty(D if conv == GT_CONV_GCN else 1 + (D + 63) // 64, dtype=torch.float32, device=dev)
        d_w = torch.empty_like(ew32) if mode in (GT_EDGE_LINEAR, GT_EDGE_TABLES) else None
        d_b = torch.empty_like(eb32) if mode == GT_EDGE_LINEAR else None
        d_dense = torch.empty_like(dense_c) if mode == GT_EDGE_DENSE else None
        ws_bytes = L.gt_aggregate_bwd_workspace_bytes(conv, mode, D, K, rows)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        toff = (C.c_int32 * max(len(ctx.tab_off), 1))(*ctx.tab_off) if ctx.tab_off else None
        _lib.launch("gt_aggregate_bwd", conv, mode, _dtype_code(h), _ptr(h), _ptr(g), N, gs.E, D, _ptr(gs.out_ptr),
                    _ptr(gs.out_dst), _ptr(gs.out_eid), _ptr(gs.deg), _ptr(gs.dis), _ptr(sp), _ptr(attr_c), K,
                    _ptr(ew32), _ptr(eb32), toff, rows, _ptr(dense_c), _ptr(dh), _ptr(d_self), _ptr(d_w), _ptr(d_b),
                    _ptr(d_dense), _ptr(ws), ws_bytes, _stream(), meta=ctx.meta)
        sdt, wdt, bdt = ctx.param_dtypes
        g_self = (d_self if conv == GT_CONV_GCN else d_self[:1]).to(sdt)
        return (dh, g_self, None if d_w is None else d_w.to(wdt), None if d_b is None else d_b.to(bdt), d_dense,
                None, None, None, None, None)


def aggregate(h, gs, conv, self_param, edge):
    """conv in {'gcn','gin'}; self_param: root_emb.weight (1,D) | eps (1,); edge: EdgeSpec."""
    cv = GT_CONV_GCN if conv == "gcn" else GT_CONV_GIN
    sp = self_param.reshape(-1)
    if edge.kind == "none":
        return _Aggregate.apply(h, sp, None, None, None, gs, cv, GT_EDGE_NONE, None, None)
    if edge.kind == "linear":
        return _Aggregate.apply(h, sp, edge.weight, edge.bias, None, gs, cv, GT_EDGE_LINEAR, edge.attr.float(), None)
    if edge.kind == "tables":
        if edge.tables is None:  # concatenated view of the embedding tables (autograd splits the gradient)
            edge.tables = torch.cat(edge.table_list, dim=0)
        return _Aggregate.apply(h, sp, edge.tables, None, None, gs, cv, GT_EDGE_TABLES, edge.attr, edge.tab_off)
    if edge.kind == "dense":
        return _Aggregate.apply(h, sp, None, None, edge.dense, gs, cv, GT_EDGE_DENSE, None, None)
    raise ValueError(edge.kind)


class _ScaleCombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, scales):
        Y = _dev(Y.float(), "Y")           # (N, T, S, F)
        N, T, S, F = Y.shape
        out = torch.empty((N, T, F), dtype=torch.float32, device=Y.device)
        _lib.launch("gt_scale_combine_fwd", _ptr(Y), _ptr(scales), N, T, S, F, _ptr(out), _stream())
        ctx.save_for_backward(scales)
        ctx.dims = (N, T, S, F)
        return out

    @staticmethod
    def backward(ctx, g):
        (scales,) = ctx.saved_tensors
        N, T, S, F = ctx.dims
        g = _dev(g.float(), "grad")
        dY = torch.empty((N, T, S, F), dtype=torch.float32, device=g.device)
        _lib.launch("gt_scale_combine_bwd", _ptr(g), _ptr(scales), N, T, S, F, _ptr(dY), _stream())
        return dY, None


def scale_combine(Y, scales):
    """sum_s Y[:, :, s] * scales[:, s] for Y (N, T, S, F) and per-node scaler factors (N, S) (PNAConv's degree
    scalers on the post-Linear's output blocks); scales carry no gradient."""
    if scales.dtype != torch.float32 or scales.shape != (Y.shape[0], Y.shape[2]) or not scales.is_contiguous():
        raise TypeError("scale_combine: contiguous fp32 (N, S) scales expected")
    return _ScaleCombine.apply(Y, scales)


# ------------------------------------------------------------------------------------------------
# per-graph segment ops (virtual node)
# ------------------------------------------------------------------------------------------------
def _bcast_add_raw(x, seg, gs):
    seg = _dev(seg, "seg")
    N, D = gs.N, seg.shape[1]
    out = torch.empty((N, D), dtype=seg.dtype, device=seg.device)
    _lib.launch("gt_segment_bcast_add", _dtype_code(seg), _ptr(x), _ptr(seg), _ptr(gs.node_graph), N, gs.B, D,
                _ptr(out), _stream())
    return out


def _segment_sum_raw(x, add, gs):
    x = _dev(x, "x")
    D = x.shape[1]
    out = torch.empty((gs.B, D), dtype=x.dtype, device=x.device)
    if gs.N >= 4096:   # load-balanced over row chunks (ragged graph sizes): needs a small workspace
        wsb = _lib.lib().gt_segment_sum_workspace_bytes(gs.N, D)
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
        _lib.launch("gt_segment_sum_ws", _dtype_code(x), _ptr(x), _ptr(add), _ptr(gs.graph_ptr), gs.N, gs.B, D, _ptr(out),
                    _ptr(ws), wsb, _stream())
    else:
        _lib.launch("gt_segment_sum", _dtype_code(x), _ptr(x), _ptr(add), _ptr(gs.graph_ptr), gs.N, gs.B, D, _ptr(out),
                    _stream())
    return out


class _BcastAdd(torch.autograd.Function):
    """h + vn[batch]  (modules/gnn_module.py:199)."""

    @staticmethod
    def forward(ctx, x, seg, gs):
        ctx.gs = gs
        return _bcast_add_raw(_dev(x, "x"), seg.to(x.dtype), gs)

    @staticmethod
    def backward(ctx, g):
        g = _dev(g, "grad")
        return g, _segment_sum_raw(g, None, ctx.gs), None


class _SegmentSum(torch.autograd.Function):
    """global_add_pool(x, batch) + add  (modules/gnn_module.py:219)."""

    @staticmethod
    def forward(ctx, x, add, gs):
        ctx.gs = gs
        ctx.has_add = add is not None
        return _segment_sum_raw(x, None if add is None else _dev(add.to(x.dtype), "add"), gs)

    @staticmethod
    def backward(ctx, g):
        g = _dev(g, "grad")
        return _bcast_add_raw(None, g, ctx.gs), (g if ctx.has_add else None), None


def segment_bcast_add(x, seg, gs):
    return _BcastAdd.apply(x, seg, gs)


def segment_sum(x, gs, add=None):
    return _SegmentSum.apply(x, add, gs)


# ------------------------------------------------------------------------------------------------
# node rows <-> token rows
# ------------------------------------------------------------------------------------------------
def _gather_raw(h, cls, gs, lay, want_mask):
    D = h.shape[1]
    tokens = (torch.empty if getattr(lay, "exact", True) else torch.zeros)((lay.rows, D), dtype=h.dtype, device=h.device)
    mask = torch.empty((lay.B, lay.max_npos), dtype=torch.bool, device=h.device) if want_mask else None
    _lib.launch("gt_seq_gather", _dtype_code(h), _ptr(h), _ptr(cls), _ptr(gs.graph_ptr), _ptr(lay.desc), lay.B,
                lay.row_stride, lay.max_npos, 1 if lay.with_cls else 0, D, _ptr(tokens), _ptr(mask), _stream())
    return tokens, mask


def _scatter_raw(tokens, base, gs, lay, want_cls):
    D = tokens.shape[1]
    h = torch.empty((gs.N, D), dtype=tokens.dtype, device=tokens.device)
    cls = torch.empty((lay.B, D), dtype=tokens.dtype, device=tokens.device) if want_cls else None
    _lib.launch("gt_seq_scatter", _dtype_code(tokens), _ptr(tokens), _ptr(base), _ptr(gs.graph_ptr),
                _ptr(gs.node_graph), _ptr(lay.desc), lay.B, lay.row_stride, 1 if lay.with_cls else 0, gs.N, D, _ptr(h),
                _ptr(cls), _stream())
    return h, cls


class _SeqGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, cls, gs, lay, want_mask):
        ctx.set_materialize_grads(False)
        h = _dev(h, "h")
        cls_c = None if cls is None else _dev(cls.reshape(-1).to(h.dtype), "cls")
        tokens, mask = _gather_raw(h, cls_c, gs, lay, want_mask)
        ctx.gs, ctx.lay = gs, lay
        ctx.cls_meta = None if cls is None else (cls.shape, cls.dtype)
        if mask is not None:
            ctx.mark_non_differentiable(mask)
        return tokens, mask

    @staticmethod
    def backward(ctx, g, _gm):
        g = _dev(g, "grad")
        dh, dcls = _scatter_raw(g, None, ctx.gs, ctx.lay, ctx.cls_meta is not None)
        if dcls is not None:
            shape, dt = ctx.cls_meta
            dcls = dcls.float().sum(0).reshape(shape).to(dt)
        return dh, dcls, None, None, None


def seq_gather(h, cls, gs, lay, want_mask=False):
    """tokens (lay.rows, d) [+ padding mask (B, max_npos), True = padding] from node rows."""
    return _SeqGather.apply(h, cls, gs, lay, want_mask)


class _SeqScatter(torch.autograd.Function):
    """unpad_batch (modules/utils.py:32-53): node rows <- token rows, truncated nodes keep `base`."""

    @staticmethod
    def forward(ctx, tokens, base, gs, lay):
        tokens = _dev(tokens, "tokens")
        base_c = None if base is None else _dev(base.to(tokens.dtype), "base")
        h, _ = _scatter_raw(tokens, base_c, gs, lay, False)
        ctx.gs, ctx.lay, ctx.has_base = gs, lay, base is not None
        return h

    @staticmethod
    def backward(ctx, g):
        g = _dev(g, "grad")
        gs, lay = ctx.gs, ctx.lay
        zero_cls = torch.zeros(g.shape[1], dtype=g.dtype, device=g.device) if lay.with_cls else None
        dtok, _ = _gather_raw(g, zero_cls, gs, lay, False)
        dbase = None
        if ctx.has_base:  # rows that were NOT overwritten keep their gradient
            keep = torch.zeros_like(g)
            kept_h, _ = _scatter_raw(torch.ones_like(dtok), None, gs, lay, False)
            dbase = g * (1 - kept_h) + keep
        return dtok, dbase, None, None


def seq_scatter(tokens, base, gs, lay):
    return _SeqScatter.apply(tokens, base, gs, lay)


# ------------------------------------------------------------------------------------------------
# fused attention
# ------------------------------------------------------------------------------------------------
class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, lay, nhead, scale, dropout_p, seed, dense_mask=None, key_valid=None, mask_value=0.0):
        qkv = _dev(qkv, "qkv")
        dense_mask = None if dense_mask is None else _dev(dense_mask.float(), "attn_mask")
        key_valid = None if key_valid is None else _dev(key_valid.float(), "valid_input_mask")
        rows, d3 = qkv.shape
        d = d3 // 3
        alloc = torch.empty if getattr(lay, "exact", True) else torch.zeros   # device-built layout: rows past the true count stay 0
        out = alloc((rows, d), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((2, nhead, rows), dtype=torch.float32, device=qkv.device)
        meta = dict(lay=lay, d=d, nhead=nhead, elt=qkv.element_size())
        _lib.launch("gt_attn_fwd", _dtype_code(qkv), _ptr(qkv), _ptr(out), _ptr(lse), rows, d, nhead, _ptr(lay.desc),
                    lay.B, lay.row_stride, lay.max_npos, _ptr(getattr(lay, "work", None)), getattr(lay, "num_work", 0),
                    _ptr(dense_mask), _ptr(key_valid), float(mask_value), scale, dropout_p, seed, _stream(), meta=meta)
        ctx.meta = meta
        ctx.save_for_backward(qkv, out, lse, dense_mask, key_valid)
        ctx.cfg = (lay, nhead, scale, dropout_p, seed, mask_value)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, out, lse, dense_mask, key_valid = ctx.saved_tensors
        lay, nhead, scale, dropout_p, seed, mask_value = ctx.cfg
        g = _dev(g.to(qkv.dtype), "grad")
        rows, d3 = qkv.shape
        # rows that belong to no sequence position do not exist in either layout -> fully written
        dqkv = torch.empty_like(qkv) if getattr(lay, "exact", True) else torch.zeros_like(qkv)
        delta = torch.empty((nhead, rows), dtype=torch.float32, device=qkv.device)
        _lib.launch("gt_attn_bwd", _dtype_code(qkv), _ptr(qkv), _ptr(out), _ptr(g), _ptr(lse), _ptr(delta),
                    _ptr(dqkv), rows, d3 // 3, nhead, _ptr(lay.desc), lay.B, lay.row_stride, lay.max_npos,
                    _ptr(getattr(lay, "work", None)), getattr(lay, "num_work", 0), _ptr(dense_mask), _ptr(key_valid),
                    float(mask_value), scale, dropout_p, seed, _stream(), meta=ctx.meta)
        return dqkv, None, None, None, None, None, None, None, None


def attention(qkv, lay, nhead, dropout_p=0.0, seed=0, scale=None, dense_mask=None, key_valid=None, mask_value=-1e6):
    """ctx rows = softmax(mask(scale q k^T)) v per (sequence, head); qkv (rows, 3*d_model).
    dense_mask (B,T,T) / key_valid (B,T): CausalSelfAttention's masked_fill(mask == 0, mask_value)."""
    d = qkv.shape[1] // 3
    if scale is None:
        scale = float(d // nhead) ** -0.5
    return _Attention.apply(qkv, lay, nhead, float(scale), float(dropout_p), int(seed), dense_mask, key_valid,
                            float(mask_value))


# ------------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------------
class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, momentum, eps, training, relu, dropout_p, seed):
        x = _dev(x, "x")
        rows, D = x.shape
        dev = x.device
        w32, b32 = _f32(weight), _f32(bias)
        y = torch.empty_like(x)
        stats = torch.empty((2, D), dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws_bytes = L.gt_batchnorm_workspace_bytes(rows, D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.launch("gt_batchnorm_fwd", _dtype_code(x), _ptr(x), _ptr(w32), _ptr(b32), _ptr(running_mean),
                    _ptr(running_var), _ptr(nbt), float(momentum), float(eps), 1 if training else 0, 1 if relu else 0,
                    None, rows, D, _ptr(y), _ptr(stats[0]), _ptr(stats[1]), float(dropout_p), int(seed), _ptr(ws), ws_bytes,
                    _stream())
        ctx.save_for_backward(x, b32, w32, stats)
        ctx.cfg = (training, relu, weight.dtype, bias.dtype, float(dropout_p), int(seed))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b32, w32, stats = ctx.saved_tensors
        training, relu, wdt, bdt, dropout_p, seed = ctx.cfg
        dy = _dev(dy.to(x.dtype), "grad")
        rows, D = x.shape
        dx = torch.empty_like(x)
        dwb = torch.empty((2, D), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ws_bytes = L.gt_batchnorm_workspace_bytes(rows, D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        _lib.launch("gt_batchnorm_bwd", _dtype_code(x), _ptr(x), _ptr(dy), _ptr(w32), _ptr(b32), _ptr(stats[0]),
                    _ptr(stats[1]), 1 if training else 0, 1 if relu else 0, rows, D, _ptr(dx), _ptr(dwb[0]), _ptr(dwb[1]),
                    dropout_p, seed, _ptr(ws), ws_bytes, _stream())
        return dx, dwb[0].to(wdt), dwb[1].to(bdt), None, None, None, None, None, None, None, None, None


def batch_norm(x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, training, relu=False,
               dropout_p=0.0, seed=0):
    """BatchNorm1d over rows of (rows, dim) with optional fused ReLU and (training) dropout (gt_batchnorm_fwd/bwd)."""
    return _BatchNorm.apply(x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, training, relu,
                            dropout_p if training else 0.0, seed)


def _dist_reduce(t, group, gather):
    """all_gather (gather=True: returns [world, ...]) or all_reduce(sum) of a small fp32 device tensor over `group`.
    RCCL takes device tensors; other backends (gloo in the tests) go through the host."""
    import torch.distributed as dist
    backend = dist.get_backend(group)
    src = t if backend == "nccl" else t.cpu()
    if gather:
        out = [torch.empty_like(src) for _ in range(dist.get_world_size(group))]
        dist.all_gather(out, src, group=group)
        res = torch.stack(out)
    else:
        dist.all_reduce(src, group=group)
        res = src
    return res.to(t.device)


class _SyncBatchNorm(torch.autograd.Function):
    """BatchNorm1d (training) whose statistics are taken over the rows of ALL data-parallel ranks: what the reference's
    single-device batch of 256 graphs sees (modules/gnn_module.py:204).  Forward: local mean / variance
    (gt_batchnorm_fwd), all-gather of (count, mean, var) = 2D + 1 floats, Chan's merge, gt_batchnorm_apply.  Backward:
    local sum(dy') and sum(dy' xhat) (gt_batchnorm_bwd), all-reduce of 2D floats, gt_batchnorm_bwd_apply."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, momentum, eps, relu, dropout_p, seed, group):
        x = _dev(x, "x")
        rows, D = x.shape
        dev = x.device
        w32, b32 = _f32(weight), _f32(bias)
        L = _lib.lib()
        ws_bytes = L.gt_batchnorm_workspace_bytes(rows, D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        local = torch.empty((2, D), dtype=torch.float32, device=dev)
        scratch = torch.empty_like(x)
        if rows > 1:   # local statistics (the apply pass of this call is discarded)
            _lib.launch("gt_batchnorm_fwd", _dtype_code(x), _ptr(x), _ptr(w32), _ptr(b32), None, None, None, float(momentum),
                        float(eps), 1, 0, None, rows, D, _ptr(scratch), _ptr(local[0]), _ptr(local[1]), 0.0, 0, _ptr(ws), ws_bytes,
                        _stream())
            mean_l, var_l = local[0], (local[1].reciprocal().square() - eps).clamp_min(0.0)
        else:
            mean_l, var_l = x[0].float() if rows == 1 else torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        packed = torch.cat([torch.full((1,), float(rows), device=dev), mean_l, var_l])
        allp = _dist_reduce(packed, group, gather=True).double()          # (world, 2D + 1)
        n_r, mean_r, var_r = allp[:, :1], allp[:, 1:D + 1], allp[:, D + 1:]
        n = n_r.sum()
        mean = (n_r * mean_r).sum(0) / n
        var = (n_r * (var_r + (mean_r - mean).square())).sum(0) / n         # biased, over all rows of all ranks
        stats = torch.stack([mean, (var + eps).rsqrt()]).float().contiguous()
        with torch.no_grad():
            if running_mean is not None:
                unbiased = var * (n / (n - 1)) if float(n) > 1 else var
                running_mean.mul_(1 - momentum).add_(momentum * mean.to(running_mean.dtype))
                running_var.mul_(1 - momentum).add_(momentum * unbiased.to(running_var.dtype))
            if nbt is not None:
                nbt.add_(1)
        y = torch.empty_like(x)
        _lib.launch("gt_batchnorm_apply", _dtype_code(x), _ptr(x), _ptr(stats[0]), _ptr(stats[1]), _ptr(w32), _ptr(b32),
                    1 if relu else 0, None, rows, D, _ptr(y), float(dropout_p), int(seed), _stream())
        ctx.save_for_backward(x, b32, w32, stats)
        ctx.cfg = (relu, weight.dtype, bias.dtype, float(dropout_p), int(seed), group, float(n))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b32, w32, stats = ctx.saved_tensors
        relu, wdt, bdt, dropout_p, seed, group, count = ctx.cfg
        dy = _dev(dy.to(x.dtype), "grad")
        rows, D = x.shape
        dev = x.device
        L = _lib.lib()
        ws_bytes = L.gt_batchnorm_workspace_bytes(rows, D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        dx = torch.empty_like(x)
        sums = torch.zeros((2, D), dtype=torch.float32, device=dev)   # [sum dy' xhat, sum dy'] over the LOCAL rows
        _lib.launch("gt_batchnorm_bwd", _dtype_code(x), _ptr(x), _ptr(dy), _ptr(w32), _ptr(b32), _ptr(stats[0]), _ptr(stats[1]), 1,
                    1 if relu else 0, rows, D, _ptr(dx), _ptr(sums[0]), _ptr(sums[1]), dropout_p, seed, _ptr(ws), ws_bytes, _stream())
        glob = _dist_reduce(sums.clone(), group, gather=False)
        _lib.launch("gt_batchnorm_bwd_apply", _dtype_code(x), _ptr(x), _ptr(dy), _ptr(w32), _ptr(b32), _ptr(stats[0]), _ptr(stats[1]),
                    _ptr(glob[1]), _ptr(glob[0]), float(count), 1 if relu else 0, rows, D, _ptr(dx), dropout_p, seed, _stream())
        # parameter gradients stay LOCAL sums: the gradient all-reduce averages them like every other parameter
        return dx, sums[0].to(wdt), sums[1].to(bdt), None, None, None, None, None, None, None, None, None


def sync_batch_norm(x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu=False, dropout_p=0.0,
                    seed=0, group=None):
    return _SyncBatchNorm.apply(x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu,
                                dropout_p, seed, group)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, resid, weight, bias, eps, dropout_p, seed):
        x = _dev(x, "x")
        resid_c = None if resid is None else _dev(resid.to(x.dtype), "resid")
        rows, D = x.shape
        w32, b32 = _f32(weight), _f32(bias)
        y = torch.empty_like(x)
        stats = torch.empty((2, rows), dtype=torch.float32, device=x.device)
        _lib.launch("gt_layernorm_fwd", _dtype_code(x), _ptr(x), _ptr(resid_c), _ptr(w32), _ptr(b32), float(eps),
                    float(dropout_p), int(seed), rows, D, _ptr(y), _ptr(stats[0]), _ptr(stats[1]), _stream())
        ctx.save_for_backward(x, resid_c, w32, stats)
        ctx.cfg = (dropout_p, seed, weight.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, resid, w32, stats = ctx.saved_tensors
        dropout_p, seed, wdt, bdt = ctx.cfg
        dy = _dev(dy.to(x.dtype), "grad")
        rows, D = x.shape
        need_x, need_r = ctx.needs_input_grad[0], resid is not None and ctx.needs_input_grad[1]
        if not (need_x or need_r):
            need_x = True
        dx = torch.empty_like(x) if need_x else None
        dres = torch.empty_like(x) if need_r else None
        dwb = torch.empty((2, D), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ws_bytes = L.gt_layernorm_bwd_workspace_bytes(rows, D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        _lib.launch("gt_layernorm_bwd", _dtype_code(x), _ptr(x), _ptr(resid), _ptr(dy), _ptr(w32), _ptr(stats[0]),
                    _ptr(stats[1]), float(dropout_p), int(seed), rows, D, _ptr(dx), _ptr(dres), _ptr(dwb[0]),
                    _ptr(dwb[1]), _ptr(ws), ws_bytes, _stream())
        return dx, dres, dwb[0].to(wdt), dwb[1].to(bdt), None, None, None


def layer_norm(x, weight, bias, eps=1e-5, resid=None, dropout_p=0.0, seed=0):
    """LayerNorm(resid + dropout(x)) over rows of (rows, dim) (gt_layernorm_fwd/bwd)."""
    return _LayerNorm.apply(x, resid, weight, bias, eps, dropout_p, seed)


# ------------------------------------------------------------------------------------------------
# nn.Linear on the matrix cores (fused bias / relu / dropout)
# ------------------------------------------------------------------------------------------------
_MATMUL_DTYPE = torch.float32


def set_matmul_dtype(dtype):
    """Compute type of fp32-STORED linears (the GNN side): torch.float32 = v_mfma_f32_16x16x4_f32
    (exact fp32, the parity mode) or torch.bfloat16 = bf16 MFMA with fp32 accumulate (operands are
    rounded to bf16 while staging; storage and master weights stay fp32)."""
    global _MATMUL_DTYPE
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError(dtype)
    _MATMUL_DTYPE = dtype


def get_matmul_dtype():
    return _MATMUL_DTYPE


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, dropout_p, seed, compute, out_dtype, ldy):
        x2 = _dev(x.reshape(-1, x.shape[-1]), "x")
        M, K = x2.shape
        N = weight.shape[0]
        ldy = N if ldy is None else int(ldy)
        w32 = _f32(weight)
        b32 = _f32(bias)
        y = torch.empty((M, ldy), dtype=out_dtype, device=x2.device)
        if act == 2:   # gelu: the backward's multiplier gelu'(z) * dropout scale is written beside y
            need = x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)
            gmul = torch.empty_like(y) if need else None
            _lib.launch("gt_linear_fwd_gelu", _dtype_code(x2), _dtype_code(y), compute, _ptr(x2), _ptr(w32), _ptr(b32), _ptr(y),
                        _ptr(gmul), M, N, K, K, ldy, float(dropout_p), int(seed), _stream())
            ctx.save_for_backward(x2, w32, gmul)
        else:
            _lib.launch("gt_linear_fwd_ld", _dtype_code(x2), _dtype_code(y), compute, _ptr(x2), _ptr(w32), _ptr(b32), _ptr(y),
                        M, N, K, ldy, act, float(dropout_p), int(seed), _stream())
            ctx.save_for_backward(x2, w32, y if act == 1 else None)
        fused = act == 1
        ctx.act = act
        ctx.cfg = (compute, dropout_p if fused else 0.0, x.shape, weight.dtype, None if bias is None else bias.dtype, ldy)
        if ldy != N:
            return y[:, :N]  # (M, N) view with row stride ldy
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w32, ymask = ctx.saved_tensors
        compute, dropout_p, xshape, wdt, bdt, ldy = ctx.cfg
        M, K = x2.shape
        N = w32.shape[0]
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], bdt is not None and ctx.needs_input_grad[2]
        ydt = ymask.dtype if ymask is not None else (dy.dtype if dy.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        if compute == GT_F32:
            ydt = torch.float32
        if ldy != N:
            dy2 = _padded_rows(dy.reshape(M, N).to(ydt), ldy)
        else:
            dy2 = _dev(dy.reshape(-1, N).to(ydt), "grad")
        dev = x2.device
        dx = torch.empty_like(x2) if need_x else None
        dw = torch.empty((N, K), dtype=torch.float32, device=dev) if (need_w or need_b) else None
        db = torch.empty(N, dtype=torch.float32, device=dev) if need_b else None
        L = _lib.lib()
        ws_bytes = L.gt_linear_bwd_workspace_bytes(compute, M, N, K)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        if ctx.act == 2:
            _lib.launch("gt_linear_bwd_mul", _dtype_code(x2), _dtype_code(dy2), compute, _ptr(x2), _ptr(w32), _ptr(dy2),
                        _ptr(ymask), None, None, _ptr(dx), _ptr(dw), _ptr(db), M, N, K, K, ldy, _ptr(ws), ws_bytes, _stream())
        else:
            _lib.launch("gt_linear_bwd_ld", _dtype_code(x2), _dtype_code(dy2), compute, _ptr(x2), _ptr(w32), _ptr(dy2),
                        _ptr(ymask), None, None, _ptr(dx), _ptr(dw), _ptr(db), M, N, K, ldy, float(dropout_p), _ptr(ws),
                        ws_bytes, _stream())
        return (None if dx is None else dx.view(xshape), None if not need_w else dw.to(wdt),
                None if db is None else db.to(bdt), None, None, None, None, None, None)


def _padded_rows(t, ld):
    """(M, N) tensor -> an (M, ld) row-padded buffer holding it with ZERO pad columns.  A gradient that
    already lives in such a buffer (the cross-entropy backward writes one) is used in place."""
    M, N = t.shape
    if (t.is_cuda and t.stride() == (ld, 1) and t.data_ptr() % 16 == 0
            and (t.storage_offset() + M * ld) * t.element_size() <= t.untyped_storage().nbytes()):
        full = t.as_strided((M, ld), (ld, 1))
        full[:, N:].zero_()
        return full
    full = torch.zeros((M, ld), dtype=t.dtype, device=t.device)
    full[:, :N].copy_(t)
    return full


def linear_supported(x, weight):
    q = 8 if x.dtype == torch.bfloat16 else 4  # 16-byte chunks of the storage type
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and weight.shape[0] % q == 0
            and weight.shape[1] % q == 0 and x.shape[-1] == weight.shape[1])


def linear(x, weight, bias=None, act=None, dropout_p=0.0, seed=0, out_dtype=None, ldy=None):
    """act in {None, 'relu', 'gelu'}; dropout_p > 0 only together with an activation (relu: mask recovered from y > 0;
    gelu (erf form): the forward writes the backward's multiplier gelu'(z) * dropout scale beside y).
    bf16-stored x always computes in bf16; fp32-stored x computes in get_matmul_dtype().
    ldy: row stride of the output buffer (multiple of 4 / 8 for fp32 / bf16) when N itself is not;
    the result is then the (M, N) column slice of an (M, ldy) buffer."""
    if x.dtype == torch.bfloat16:
        compute = GT_BF16
    else:
        compute = GT_BF16 if _MATMUL_DTYPE == torch.bfloat16 else GT_F32
    if out_dtype is None:
        out_dtype = x.dtype
    if compute == GT_F32 and out_dtype != torch.float32:
        raise ValueError("fp32 compute writes fp32")
    if act not in (None, "relu", "gelu"):
        raise ValueError(act)
    a = {None: 0, "relu": 1, "gelu": 2}[act]
    if dropout_p > 0 and a == 0:
        raise ValueError("fused dropout needs a fused activation")
    return _Linear.apply(x, weight, bias, a, float(dropout_p), int(seed), compute, out_dtype, ldy)


class _PadCols(torch.autograd.Function):
    """(rows, K) -> (rows, Kp >= K), zero filled; the adjoint drops the padding columns (gt_repitch)."""

    @staticmethod
    def forward(ctx, x, cols):
        x = _dev(x, "x")
        out = torch.empty((x.shape[0], cols), dtype=x.dtype, device=x.device)
        _lib.launch("gt_repitch", _ptr(out), cols, _ptr(x), x.shape[1], x.shape[0], x.element_size(), _stream())
        ctx.k = x.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        g = _dev(g, "grad")
        out = torch.empty((g.shape[0], ctx.k), dtype=g.dtype, device=g.device)
        _lib.launch("gt_repitch", _ptr(out), ctx.k, _ptr(g), g.shape[1], g.shape[0], g.element_size(), _stream())
        return out, None


def pad_cols(x, cols):
    return x if x.shape[1] == cols else _PadCols.apply(x, cols)


def linear_module(mod, x, act=None, dropout_p=0.0, seed=0):
    """Apply an nn.Linear through the HIP GEMM.  The kernel moves 16-byte chunks: an output width N that is
    not a multiple of 4 (fp32) / 8 (bf16) is written into row-padded storage; an input width K that is not
    (the 37-feature TU node encoder, dataset/tud.py:65) gets x and W zero-padded along K, which leaves every
    product unchanged."""
    if x.dim() != 2 or not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16) or x.shape[-1] != mod.weight.shape[1]:
        raise RuntimeError("graphtrans_amd.ops.linear_module: expected a 2-D fp32/bf16 GPU tensor of %d features, got %s %s on %s "
                           "(no CPU / eager fallback)" % (mod.weight.shape[1], tuple(x.shape), x.dtype, x.device))
    q = 8 if x.dtype == torch.bfloat16 else 4
    N, K = mod.weight.shape
    w = mod.weight
    if K % q:
        Kp = (K + q - 1) // q * q
        x, w = pad_cols(x.contiguous(), Kp), pad_cols(w, Kp)
    return linear(x, w, mod.bias, act=act, dropout_p=dropout_p, seed=seed, ldy=None if N % q == 0 else (N + q - 1) // q * q)


# ------------------------------------------------------------------------------------------------
# PNA multi-aggregator message passing
# ------------------------------------------------------------------------------------------------
class _PnaAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, V, gs, towers):
        U, V = _dev(U.float(), "U"), _dev(V.float(), "V")
        N, D = V.shape
        out = torch.empty((N, towers, 4 * (D // towers)), dtype=torch.float32, device=V.device)
        mean_v = torch.empty_like(V)
        arg = torch.empty((N, 2, D), dtype=torch.int32, device=V.device)
        _lib.launch("gt_pna_aggregate_fwd", _ptr(U), _ptr(V), N, D, towers, _ptr(gs.in_ptr), _ptr(gs.in_src),
                    _ptr(gs.in_eid), _ptr(out), _ptr(mean_v), _ptr(arg), _stream())
        ctx.save_for_backward(V, out, mean_v, arg)
        ctx.gs, ctx.towers = gs, towers
        return out

    @staticmethod
    def backward(ctx, g):
        V, out, mean_v, arg = ctx.saved_tensors
        gs = ctx.gs
        g = _dev(g.float(), "grad")
        N, D = V.shape
        dU, dV = torch.empty_like(V), torch.empty_like(V)
        _lib.launch("gt_pna_aggregate_bwd", _ptr(V), _ptr(out), _ptr(mean_v), _ptr(arg), _ptr(g), N, D, ctx.towers,
                    _ptr(gs.in_ptr), _ptr(gs.out_ptr), _ptr(gs.out_dst), _ptr(gs.out_eid), _ptr(dU), _ptr(dV), _stream())
        return dU, dV, None, None


def pna_aggregate(U, V, gs, towers):
    """(N, towers, 4F): [U+mean V | U+max V | U+min V | std V] over the in-edges (gt_pna_aggregate_*)."""
    return _PnaAggregate.apply(U, V, gs, towers)


EMBED_SORT_MAX_ROWS = 16384   # gt_embed_sort's per-block LDS histogram


class _EmbedSum(torch.autograd.Function):
    """sum_t table_t[min(idx_t, clamp_t)] (gt_embed_sum_fwd / _bwd).  `cols` is a list of
    (int64 tensor, element offset, element stride, clamp) column descriptors."""

    @staticmethod
    def forward(ctx, cols, *tables):
        T = len(tables)
        tables = [_dev(t, "table") for t in tables]
        if any(t.dtype != torch.float32 for t in tables):
            raise TypeError("embed_sum: fp32 tables only")
        D = tables[0].shape[1]
        N = cols[0][0].shape[0]
        I64, P = C.c_int64 * T, C.c_void_p * T
        idx = P(*[c[0].data_ptr() + 8 * c[1] for c in cols])
        strides = I64(*[c[2] for c in cols])
        clamp = I64(*[c[3] for c in cols])
        tabs = P(*[t.data_ptr() for t in tables])
        out = torch.empty((N, D), dtype=torch.float32, device=tables[0].device)
        _lib.launch("gt_embed_sum_fwd", T, idx, strides, clamp, tabs, N, D, _ptr(out), _stream())
        rows = [t.shape[0] for t in tables]
        ctx.cols, ctx.meta = cols, (T, N, D, rows, tables[0].device)
        ctx.desc = (idx, strides, clamp)
        ctx.plan = None
        if any(t.requires_grad for t in tables) and max(rows) <= EMBED_SORT_MAX_ROWS:
            # the backward sums each table row's gradient rows in node order: sort the node ids by row now
            L = _lib.lib()
            rows_c = I64(*rows)
            plan = torch.empty(L.gt_embed_sort_plan_bytes(T, rows_c, N), dtype=torch.uint8, device=out.device)
            wsb = L.gt_embed_sort_workspace_bytes(T, rows_c, N)
            ws = torch.empty(wsb, dtype=torch.uint8, device=out.device)
            _lib.launch("gt_embed_sort", T, idx, strides, clamp, rows_c, N, _ptr(plan), plan.numel(), _ptr(ws), wsb, _stream())
            ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        T, N, D, rows, device = ctx.meta
        idx, strides, clamp = ctx.desc
        g = _dev(g.float(), "grad")
        I64, P = C.c_int64 * T, C.c_void_p * T
        rows_c = I64(*rows)
        grads = [torch.empty((r, D), dtype=torch.float32, device=device) if ctx.needs_input_grad[1 + t] else None
                 for t, r in enumerate(rows)]
        dt = P(*[(x.data_ptr() if x is not None else None) for x in grads])
        if ctx.plan is not None:
            ws_bytes = _lib.lib().gt_embed_sum_bwd_sorted_workspace_bytes(T, N, D)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            _lib.launch("gt_embed_sum_bwd_sorted", T, rows_c, _ptr(g), N, D, _ptr(ctx.plan), dt, _ptr(ws), ws_bytes, _stream())
            return (None, *grads)
        ws_bytes = _lib.lib().gt_embed_sum_bwd_workspace_bytes(T, rows_c, D)   # huge tables: fixed-point atomics
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        _lib.launch("gt_embed_sum_bwd", T, idx, strides, clamp, rows_c, _ptr(g), N, D, dt, _ptr(ws), ws_bytes, _stream())
        return (None, *grads)


def embed_sum(columns, tables, clamps=None):
    """out[n] = sum_t tables[t][min(columns[t][n], clamps[t])].  `columns[t]` is an int64 device
    tensor of shape (N,), possibly a strided view of an (N, C) matrix (no copy is made)."""
    cols = []
    for t, c in enumerate(columns):
        if c.dtype != torch.int64 or c.dim() != 1:
            raise TypeError("embed_sum: index columns must be 1-D int64")
        if not c.is_cuda:
            raise RuntimeError("index must be a GPU tensor: graphtrans_amd has no CPU fallback")
        clamp = -1 if clamps is None or clamps[t] is None else int(clamps[t])
        cols.append((c, 0, c.stride(0) if c.shape[0] > 1 else 1, clamp))
    return _EmbedSum.apply(cols, *tables)


# ------------------------------------------------------------------------------------------------
# softmax cross-entropy over the stacked prediction heads
# ------------------------------------------------------------------------------------------------
class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _dev(x.contiguous(), "x")
        if x.dtype not in (torch.float32, torch.bfloat16) or x.numel() % 4:
            raise TypeError("dropout: fp32 / bf16 tensor with a multiple of 4 elements expected")
        y = torch.empty_like(x)
        _lib.launch("gt_dropout", _dtype_code(x), _ptr(x), _ptr(y), x.numel(), float(p), int(seed), _stream())
        ctx.cfg = (float(p), int(seed))
        return y

    @staticmethod
    def backward(ctx, g):
        g = _dev(g.contiguous(), "grad")
        out = torch.empty_like(g)
        _lib.launch("gt_dropout", _dtype_code(g), _ptr(g), _ptr(out), g.numel(), ctx.cfg[0], ctx.cfg[1], _stream())
        return out, None, None


def dropout(x, p, training=True, seed=None):
    """F.dropout(x, p, training) on gt_dropout (counter-hash mask, replayed by the backward); identity when not training."""
    if not training or p <= 0.0:
        return x
    if seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
    return _Dropout.apply(x, p, seed)


_VALIDATE = os.environ.get("GT_VALIDATE", "0") not in ("", "0")


def set_validate(on):
    """validate mode: index-like inputs of the losses are checked on the device and errors raised on the host (costs a
    device -> host sync per call); off by default.  Also GT_VALIDATE=1."""
    global _VALIDATE
    _VALIDATE = bool(on)


class _Xent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stacked, target):
        B, L, C = stacked.shape
        if stacked.dtype != torch.float32 or stacked.stride(2) != 1 or stacked.stride(1) != C:
            raise TypeError("softmax_xent: fp32 (B, L, C) logits with contiguous heads expected")
        if not stacked.is_cuda:
            raise RuntimeError("logits must be a GPU tensor: graphtrans_amd has no CPU fallback")
        ld = stacked.stride(0) if B > 1 else L * C
        target = target.contiguous()
        dev = stacked.device
        aux = torch.empty(2 * B * L + L + 2, dtype=torch.float32, device=dev)   # ..., loss, status
        lse, row_loss, head_scale, loss = aux[:B * L], aux[B * L:2 * B * L], aux[2 * B * L:2 * B * L + L], aux[2 * B * L + L:]
        _lib.launch("gt_xent_fwd", _ptr(stacked), B, L, C, ld, _ptr(target), target.stride(0), _ptr(lse), _ptr(row_loss),
                    _ptr(head_scale), _ptr(loss), _stream())
        ctx.save_for_backward(stacked, target, lse, head_scale)
        ctx.ld = ld
        if _VALIDATE:   # device -> host read of the status word (a sync): out-of-range class indices, like torch's assert
            bad = int(loss[1].item())
            if bad:
                raise IndexError("softmax_xent: %d target(s) outside [0, %d) and not ignore_index (-100)" % (bad, C))
        return loss[0].reshape(())

    @staticmethod
    def backward(ctx, g):
        stacked, target, lse, head_scale = ctx.saved_tensors
        B, L, C = stacked.shape
        ld = ctx.ld
        g = g.to(torch.float32).contiguous()
        buf = torch.empty((B, ld), dtype=torch.float32, device=stacked.device)
        _lib.launch("gt_xent_bwd", _ptr(stacked), _ptr(lse), _ptr(head_scale), _ptr(target), target.stride(0), _ptr(g),
                    B, L, C, ld, _ptr(buf), _stream())
        return buf[:, :L * C].view(B, L, C), None


def softmax_xent(stacked, target):
    """mean_l CrossEntropyLoss()(stacked[:, l], target[:, l]) for (B, L, C) fp32 logits whose batch rows
    may be padded (stride(0) >= L*C); ignore_index -100 like torch's default (gt_xent_fwd / _bwd)."""
    if target.dtype != torch.int64 or target.dim() != 2 or target.shape[1] < stacked.shape[1]:
        raise TypeError("softmax_xent: int64 (B, >=L) targets expected")
    return _Xent.apply(stacked, target)


class _MaskedBce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, y, den):
        B, T = pred.shape
        if pred.dtype != torch.float32 or pred.stride(1) != 1 or y.dtype != torch.float32 or y.stride(1) != 1:
            raise TypeError("masked_bce: fp32 (B, T) logits and targets with contiguous rows expected")
        if not pred.is_cuda:
            raise RuntimeError("logits must be a GPU tensor: graphtrans_amd has no CPU fallback")
        ld = pred.stride(0) if B > 1 else T
        tld = y.stride(0) if B > 1 else T
        aux = torch.empty(2 * B + 2, dtype=torch.float32, device=pred.device)
        _lib.launch("gt_bce_masked_fwd", _ptr(pred), _ptr(y), B, T, ld, tld, _ptr(den), _ptr(aux), aux.data_ptr() + 4 * B,
                    aux.data_ptr() + 8 * B, _stream())
        ctx.save_for_backward(pred, y, aux)
        ctx.ld, ctx.tld = ld, tld
        return aux[2 * B].reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, y, aux = ctx.saved_tensors
        B, T = pred.shape
        g = g.to(torch.float32).contiguous()
        buf = torch.empty((B, ctx.ld), dtype=torch.float32, device=pred.device)
        _lib.launch("gt_bce_masked_bwd", _ptr(pred), _ptr(y), aux.data_ptr() + 8 * B, _ptr(g), B, T, ctx.ld, ctx.tld, _ptr(buf),
                    _stream())
        return buf[:, :T], None, None


def masked_bce(pred, y, den=None):
    """BCEWithLogitsLoss over the labelled (non-NaN) entries of y (dataset/mol.py:24-31) in two launches, backward in
    one; `den` (1-element fp32 GPU tensor) replaces the local labelled count as the denominator."""
    if pred.dim() != 2 or pred.shape != y.shape:
        raise TypeError("masked_bce: (B, T) logits and targets of the same shape expected")
    return _MaskedBce.apply(pred, y, den)


# ------------------------------------------------------------------------------------------------
# per-tower linears (PNAConv's pre_nns / post_nns): column slices in, column slices out
# ------------------------------------------------------------------------------------------------
class _TowerLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, compute):
        x = _dev(x, "x")                      # (M, T, K) contiguous: tower t is the column slice [t*K, (t+1)*K)
        M, T, K = x.shape
        Nout = weight.shape[1]
        w32 = _f32(weight)                    # (T, Nout, K)
        b32 = _f32(bias)                      # (T, Nout) or None
        y = torch.empty((M, T, Nout), dtype=x.dtype, device=x.device)
        w32 = w32.contiguous()
        b32 = None if b32 is None else b32.contiguous()
        # one grouped launch: group t = column slice [t*K, (t+1)*K) of x -> [t*Nout, (t+1)*Nout) of y
        _lib.launch("gt_linear_fwd_grouped", _dtype_code(x), _dtype_code(y), compute, _ptr(x), _ptr(w32), _ptr(b32), _ptr(y),
                    M, Nout, K, T * K, T * Nout, T, K, Nout, 0, 0.0, 0, _stream())
        ctx.save_for_backward(x, w32)
        ctx.cfg = (compute, weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w32 = ctx.saved_tensors
        compute, wdt, bdt = ctx.cfg
        M, T, K = x.shape
        Nout = w32.shape[1]
        dy = _dev(dy.to(x.dtype), "grad")
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or (bdt is not None and ctx.needs_input_grad[2])
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty((T, Nout, K), dtype=torch.float32, device=x.device) if need_w else None
        db = torch.empty((T, Nout), dtype=torch.float32, device=x.device) if (need_w and bdt is not None) else None
        ws_bytes = _lib.lib().gt_linear_bwd_grouped_workspace_bytes(compute, M, Nout, K, T)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
        _lib.launch("gt_linear_bwd_grouped", _dtype_code(x), _dtype_code(dy), compute, _ptr(x), _ptr(w32), _ptr(dy), None, None, None,
                    _ptr(dx), _ptr(dw), _ptr(db), M, Nout, K, T * K, T * Nout, T, K, Nout, 0.0, _ptr(ws), ws_bytes, _stream())
        return dx, (None if dw is None else dw.to(wdt)), (None if db is None else db.to(bdt)), None


def tower_linear(x, weight, bias=None):
    """y[:, t] = x[:, t] @ weight[t].T + bias[t] for (M, T, K) x, (T, Nout, K) weight -> (M, T, Nout): one grouped
    launch (gt_linear_*_grouped, grid.y = tower) on column slices (no transposes, no per-tower copies).  K and Nout multiples of 4 (8 for bf16)."""
    if x.dim() != 3 or weight.dim() != 3 or x.shape[1] != weight.shape[0] or x.shape[2] != weight.shape[2]:
        raise ValueError("tower_linear: x (M, T, K), weight (T, Nout, K)")
    compute = GT_BF16 if (x.dtype == torch.bfloat16 or _MATMUL_DTYPE == torch.bfloat16) else GT_F32
    return _TowerLinear.apply(x, weight, bias, compute)
