"""Composite layer ops (gt_encoder_layer_*, gt_gcn_layer_*, gt_vn_update_*): one autograd node and one
C call per layer and direction.  The C side (csrc/layers.hip) sequences the same kernels the
fine-grained ops in ops.py launch one by one; this module only builds the descriptor structs and owns
the saved-activation / gradient buffers.  Used automatically by the modules when a layer's
configuration is covered (see `*_eligible`); otherwise the fine-grained path runs.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import GT_BF16, GT_EDGE_LINEAR, GT_EDGE_NONE, GT_EDGE_TABLES, GT_F32
from .graph import _ptr, _stream

_fp = C.c_void_p


class EncoderLayerDesc(C.Structure):
    _fields_ = [("rows", C.c_int64), ("d_model", C.c_int64), ("ffn", C.c_int64),
                ("nhead", C.c_int32), ("dtype", C.c_int32), ("compute", C.c_int32), ("training", C.c_int32),
                ("seq_desc", _fp), ("num_seqs", C.c_int64), ("row_stride", C.c_int64), ("max_npos", C.c_int64),
                ("work_items", _fp), ("num_work", C.c_int64), ("dropout_p", C.c_float), ("ln_eps", C.c_float),
                ("act", C.c_int32), ("reserved_", C.c_int32), ("seed", C.c_uint64)] + \
               [(n, _fp) for n in ("in_w", "in_b", "out_w", "out_b", "l1_w", "l1_b", "l2_w", "l2_b", "n1_w", "n1_b",
                                   "n2_w", "n2_b")]


class GcnLayerDesc(C.Structure):
    _fields_ = [("N", C.c_int64), ("E", C.c_int64), ("B", C.c_int64), ("D", C.c_int64),
                ("edge_mode", C.c_int32), ("has_vn", C.c_int32), ("relu", C.c_int32), ("residual", C.c_int32),
                ("training", C.c_int32), ("compute", C.c_int32),
                ("edge_cols", C.c_int64), ("table_rows", C.c_int64), ("tab_off", C.c_int32 * 4),
                ("bn_momentum", C.c_float), ("bn_eps", C.c_float)] + \
               [(n, _fp) for n in ("graph_ptr", "node_graph", "in_ptr", "in_src", "in_eid", "out_ptr", "out_dst", "out_eid",
                                   "deg", "dis", "edge_attr", "lin_w", "lin_b", "root", "edge_w", "edge_b", "bn_w", "bn_b",
                                   "bn_rm", "bn_rv", "bn_nbt", "ev_x_ready", "ev_dx_wait")] + \
               [("seed", C.c_uint64), ("dropout_p", C.c_float), ("x_has_vn", C.c_int32), ("vn_next", _fp), ("ev_vn_next", _fp),
                ("lin_wt", _fp), ("prev_saved", _fp), ("prev_bn_w", _fp), ("prev_bn_b", _fp), ("prev_bn_part", _fp),
                ("bn_part_in", _fp), ("prev_relu", C.c_int32), ("bn_nparts_in", C.c_int32), ("ev_graph_ready", _fp),
                ("dx_bcast", _fp), ("dx_bcast_idx", _fp)]


class GinLayerDesc(C.Structure):
    _fields_ = [("N", C.c_int64), ("E", C.c_int64), ("B", C.c_int64), ("D", C.c_int64),
                ("edge_mode", C.c_int32), ("has_vn", C.c_int32), ("relu", C.c_int32), ("residual", C.c_int32),
                ("training", C.c_int32), ("compute", C.c_int32),
                ("edge_cols", C.c_int64), ("table_rows", C.c_int64), ("tab_off", C.c_int32 * 4),
                ("bn_momentum", C.c_float), ("bn_eps", C.c_float)] + \
               [(n, _fp) for n in ("graph_ptr", "node_graph", "in_ptr", "in_src", "in_eid", "out_ptr", "out_dst", "out_eid",
                                   "edge_attr", "eps", "edge_w", "edge_b", "w1", "b1", "bn1_w", "bn1_b", "w2", "b2", "bn_w",
                                   "bn_b", "bn1_rm", "bn1_rv", "bn_rm", "bn_rv", "bn1_nbt", "bn_nbt", "ev_x_ready",
                                   "ev_dx_wait")] + \
               [("seed", C.c_uint64), ("dropout_p", C.c_float), ("x_has_vn", C.c_int32), ("vn_next", _fp), ("ev_vn_next", _fp),
                ("w1_t", _fp), ("w2_t", _fp)]


class VnUpdateDesc(C.Structure):
    _fields_ = [("N", C.c_int64), ("B", C.c_int64), ("D", C.c_int64),
                ("residual", C.c_int32), ("training", C.c_int32), ("compute", C.c_int32), ("pad_", C.c_int32),
                ("bn_momentum", C.c_float), ("bn_eps", C.c_float)] + \
               [(n, _fp) for n in ("graph_ptr", "node_graph", "identity_graph", "w1", "b1", "bn1_w", "bn1_b", "w2", "b2",
                                   "bn2_w", "bn2_b", "bn1_rm", "bn1_rv", "bn2_rm", "bn2_rv", "bn1_nbt", "bn2_nbt")] + \
               [("seed", C.c_uint64), ("dropout_p", C.c_float), ("pad2_", C.c_int32), ("ev_dx_done", _fp)]


class PnaLayerDesc(C.Structure):   # gt_pna_layer
    _fields_ = [("N", C.c_int64), ("E", C.c_int64), ("D", C.c_int64),
                ("T", C.c_int32), ("S", C.c_int32), ("training", C.c_int32), ("compute", C.c_int32),
                ("bn_momentum", C.c_float), ("bn_eps", C.c_float)] + \
               [(n, _fp) for n in ("in_ptr", "in_src", "in_eid", "out_ptr", "out_dst", "out_eid", "scales", "pre_w", "pre_b", "post_w",
                                   "post_b", "lin_w", "lin_b", "bn_w", "bn_b", "bn_rm", "bn_rv", "bn_nbt", "d_pre_w", "d_pre_b",
                                   "d_post_w", "d_post_b")] + \
               [("seed", C.c_uint64), ("dropout_p", C.c_float), ("pad_", C.c_int32)]


def _any_sync(*bns):
    from .modules.norm import any_sync
    return any_sync(*bns)


def _bind():
    return _lib.lib()  # signatures are declared in _lib.SIGNATURES (descriptor pointers as void*)


def _bytes(n, dev):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=dev)


def _split(flat, params):
    out, off = [], 0
    for p in params:
        n = p.numel()
        out.append(flat[off:off + n].view(p.shape))
        off += n
    return out


def _compute_code():
    from . import ops
    return GT_BF16 if ops.get_matmul_dtype() == torch.bfloat16 else GT_F32


def _f32c(p):
    if p.dtype != torch.float32 or not p.is_contiguous():
        raise TypeError("composite layers need contiguous fp32 parameters")
    return p


# ------------------------------------------------------------------------------------------------
# encoder layer
# ------------------------------------------------------------------------------------------------
ENC_PARAM_ORDER = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                   "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                   "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


def encoder_layer_params(mod):
    sa = mod.self_attn
    return [sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, mod.linear1.weight, mod.linear1.bias,
            mod.linear2.weight, mod.linear2.bias, mod.norm1.weight, mod.norm1.bias, mod.norm2.weight, mod.norm2.bias]


class _EncoderLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lay, nhead, dropout_p, seed, training, ln_eps, act, *params):
        L = _bind()
        x = x.contiguous()
        rows, d = x.shape
        desc = EncoderLayerDesc()
        desc.rows, desc.d_model, desc.ffn = rows, d, params[4].shape[0]
        desc.nhead = nhead
        desc.dtype = GT_BF16 if x.dtype == torch.bfloat16 else GT_F32
        desc.compute = _compute_code()
        desc.training = 1 if training else 0
        desc.seq_desc = _ptr(lay.desc)
        desc.num_seqs, desc.row_stride, desc.max_npos = lay.B, lay.row_stride, lay.max_npos
        desc.work_items, desc.num_work = _ptr(getattr(lay, "work", None)), getattr(lay, "num_work", 0)
        desc.dropout_p, desc.ln_eps, desc.seed = float(dropout_p), float(ln_eps), int(seed)
        desc.act = ENC_ACT[act]
        for name, p in zip(("in_w", "in_b", "out_w", "out_b", "l1_w", "l1_b", "l2_w", "l2_b", "n1_w", "n1_b", "n2_w",
                            "n2_b"), params):
            setattr(desc, name, _ptr(_f32c(p)))
        saved = _bytes(L.gt_encoder_layer_saved_bytes(C.byref(desc)), x.device)
        if not getattr(lay, "exact", True):
            saved.zero_()   # device-built layout (upper-bound row count): the attention kernels skip the rows past the true count
        y = torch.empty_like(x)
        _lib.check(L.gt_encoder_layer_fwd(C.byref(desc), _ptr(x), _ptr(y), _ptr(saved), _stream()), "gt_encoder_layer_fwd")
        ctx.save_for_backward(x, saved, *params)
        ctx.desc, ctx.lay = desc, lay
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _bind()
        x, saved = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        desc = ctx.desc
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = torch.empty_like(x)
        grads = torch.empty(L.gt_encoder_layer_grad_elems(C.byref(desc)), dtype=torch.float32, device=x.device)
        ws_bytes = L.gt_encoder_layer_workspace_bytes(C.byref(desc))
        ws = _bytes(ws_bytes, x.device)
        if not getattr(ctx.lay, "exact", True):
            ws.zero_()
        _lib.check(L.gt_encoder_layer_bwd(C.byref(desc), _ptr(x), _ptr(dy), _ptr(saved), _ptr(dx), _ptr(grads), _ptr(ws),
                                          ws_bytes, _stream()), "gt_encoder_layer_bwd")
        return (dx, None, None, None, None, None, None, None, *_split(grads, params))


class _EncoderLayerPooled(torch.autograd.Function):
    """The LAST encoder layer under cls / last pooling (gt_encoder_layer_pooled_*): x (rows, d) -> y (B, d), the layer's output in the
    pooled row of every sequence only (what transformer_out[-1] reads, models/gnn_transformer.py:113-114)."""

    @staticmethod
    def forward(ctx, x, lay, nhead, dropout_p, seed, training, ln_eps, act, *params):
        L = _bind()
        x = x.contiguous()
        rows, d = x.shape
        desc = EncoderLayerDesc()
        desc.rows, desc.d_model, desc.ffn = rows, d, params[4].shape[0]
        desc.nhead = nhead
        desc.dtype = GT_BF16 if x.dtype == torch.bfloat16 else GT_F32
        desc.compute = _compute_code()
        desc.training = 1 if training else 0
        desc.seq_desc = _ptr(lay.desc)
        desc.num_seqs, desc.row_stride, desc.max_npos = lay.B, lay.row_stride, lay.max_npos
        desc.work_items, desc.num_work = _ptr(getattr(lay, "work", None)), getattr(lay, "num_work", 0)
        desc.dropout_p, desc.ln_eps, desc.seed = float(dropout_p), float(ln_eps), int(seed)
        desc.act = ENC_ACT[act]
        for name, p in zip(("in_w", "in_b", "out_w", "out_b", "l1_w", "l1_b", "l2_w", "l2_b", "n1_w", "n1_b", "n2_w",
                            "n2_b"), params):
            setattr(desc, name, _ptr(_f32c(p)))
        zero = not getattr(lay, "exact", True)
        saved = _bytes(L.gt_encoder_layer_pooled_saved_bytes(C.byref(desc)), x.device)
        if zero:
            saved.zero_()
        y = torch.empty((lay.B, d), dtype=x.dtype, device=x.device)
        _lib.check(L.gt_encoder_layer_pooled_fwd(C.byref(desc), _ptr(x), _ptr(lay.last_rows), _ptr(y), _ptr(saved), _stream()),
                   "gt_encoder_layer_pooled_fwd")
        ctx.save_for_backward(x, saved, *params)
        ctx.desc, ctx.lay = desc, lay
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _bind()
        x, saved = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        desc = ctx.desc
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = torch.empty_like(x)
        grads = torch.empty(L.gt_encoder_layer_grad_elems(C.byref(desc)), dtype=torch.float32, device=x.device)
        ws_bytes = L.gt_encoder_layer_pooled_workspace_bytes(C.byref(desc))
        ws = _bytes(ws_bytes, x.device)
        if not getattr(ctx.lay, "exact", True):
            ws.zero_()
        _lib.check(L.gt_encoder_layer_pooled_bwd(C.byref(desc), _ptr(x), _ptr(ctx.lay.last_rows), _ptr(dy), _ptr(saved), _ptr(dx), _ptr(grads),
                                                 _ptr(ws), ws_bytes, _stream()), "gt_encoder_layer_pooled_bwd")
        return (dx, None, None, None, None, None, None, None, *_split(grads, params))


def encoder_layer_pooled(x, mod, lay, nhead, dropout_p, seed, training, activation="relu"):
    return _EncoderLayerPooled.apply(x, lay, nhead, dropout_p, seed, training, mod.norm1.eps, activation, *encoder_layer_params(mod))


ENC_ACT = {"relu": 0, "gelu": 1}   # gt_encoder_layer.act


def encoder_layer_eligible(mod, x, activation):
    return (activation in ENC_ACT and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[1] % 8 == 0
            and mod.linear1.weight.shape[0] % 8 == 0 and all(p.dtype == torch.float32 for p in mod.parameters()))


def encoder_layer(x, mod, lay, nhead, dropout_p, seed, training, activation="relu"):
    return _EncoderLayer.apply(x, lay, nhead, dropout_p, seed, training, mod.norm1.eps, activation, *encoder_layer_params(mod))


# ------------------------------------------------------------------------------------------------
# GCN layer (+ virtual-node add, BatchNorm, ReLU, residual)
# ------------------------------------------------------------------------------------------------
def _fill_graph(desc, gs):
    for name in ("graph_ptr", "node_graph", "in_ptr", "in_src", "in_eid", "out_ptr", "out_dst", "out_eid", "deg", "dis"):
        if hasattr(desc, name):
            setattr(desc, name, _ptr(getattr(gs, name)))


class _GcnLayer(torch.autograd.Function):
    """(x, y) = layer(h_in, vn): x = h_in + vn[batch] (the reference's updated h_list[layer]),
    y = BN(GCNConv(x)) [relu] [+ x]."""

    @staticmethod
    def forward(ctx, h_in, vn, gs, spec, relu, residual, training, bn, dropout_p, seed, *params):
        ctx.set_materialize_grads(False)  # an unused output (x without consumers) must not cost a zero fill
        L = _bind()
        h_in = h_in.contiguous()
        N, D = h_in.shape
        desc = GcnLayerDesc()
        desc.N, desc.E, desc.B, desc.D = N, gs.E, gs.B, D
        desc.has_vn = 0 if vn is None else 1
        desc.relu, desc.residual, desc.training = int(relu), int(residual), int(training)
        desc.compute = _compute_code()
        desc.bn_momentum, desc.bn_eps = float(bn.momentum), float(bn.eps)
        desc.dropout_p, desc.seed = (float(dropout_p) if training else 0.0), int(seed)
        _fill_graph(desc, gs)
        lin_w, lin_b, root, bn_w, bn_b = params[:5]
        edge_params = params[5:]
        keep = []
        if spec.kind == "linear":
            desc.edge_mode = GT_EDGE_LINEAR
            attr = spec.attr.float().contiguous()
            desc.edge_cols = attr.shape[1]
            desc.edge_attr, desc.edge_w, desc.edge_b = _ptr(attr), _ptr(_f32c(edge_params[0])), _ptr(_f32c(edge_params[1]))
            keep.append(attr)
        elif spec.kind == "tables":
            desc.edge_mode = GT_EDGE_TABLES
            attr = spec.attr.contiguous()
            tables = torch.cat([t for t in edge_params], dim=0)
            desc.edge_cols, desc.table_rows = attr.shape[1], tables.shape[0]
            for i, o in enumerate(spec.tab_off):
                desc.tab_off[i] = o
            desc.edge_attr, desc.edge_w = _ptr(attr), _ptr(tables)
            keep += [attr, tables]
        else:
            desc.edge_mode = GT_EDGE_NONE
        desc.lin_w, desc.lin_b, desc.root = _ptr(_f32c(lin_w)), _ptr(_f32c(lin_b)), _ptr(_f32c(root))
        desc.bn_w, desc.bn_b = _ptr(_f32c(bn_w)), _ptr(_f32c(bn_b))
        desc.bn_rm, desc.bn_rv, desc.bn_nbt = _ptr(bn.running_mean), _ptr(bn.running_var), _ptr(bn.num_batches_tracked)
        dev = h_in.device
        saved = _bytes(L.gt_gcn_layer_saved_bytes(C.byref(desc)), dev)
        ws_bytes = L.gt_gcn_layer_workspace_bytes(C.byref(desc))
        ws = _bytes(ws_bytes, dev)
        y = torch.empty_like(h_in)
        vn_c = None if vn is None else vn.contiguous()
        x = torch.empty_like(h_in) if vn is not None else h_in
        _lib.check(L.gt_gcn_layer_fwd(C.byref(desc), _ptr(h_in), _ptr(vn_c), _ptr(x) if vn is not None else None, _ptr(y),
                                      _ptr(saved), _ptr(ws), ws_bytes, _stream()), "gt_gcn_layer_fwd")
        ctx.save_for_backward(x, saved, *params)
        ctx.desc, ctx.keep, ctx.spec, ctx.has_vn = desc, keep, spec, vn is not None
        # without a virtual node x IS h_in: hand back a placeholder instead of aliasing an input
        return (x if vn is not None else h_in.new_empty(0)), y

    @staticmethod
    def backward(ctx, gx, gy):
        L = _bind()
        x, saved = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        desc = ctx.desc
        dev = x.device
        if gy is None:  # only x had consumers: the layer itself contributes nothing
            gy = torch.zeros_like(x)
        gy = gy.contiguous()
        gx_c = gx.contiguous() if (gx is not None and ctx.has_vn) else None
        d_h = torch.empty_like(x)
        d_vn = torch.empty((desc.B, desc.D), dtype=torch.float32, device=dev) if ctx.has_vn else None
        grads = torch.empty(L.gt_gcn_layer_grad_elems(C.byref(desc)), dtype=torch.float32, device=dev)
        ws_bytes = L.gt_gcn_layer_workspace_bytes(C.byref(desc))
        ws = _bytes(ws_bytes, dev)
        _lib.check(L.gt_gcn_layer_bwd(C.byref(desc), _ptr(x), _ptr(gy), _ptr(gx_c), _ptr(saved), _ptr(d_h), _ptr(d_vn),
                                      _ptr(grads), _ptr(ws), ws_bytes, _stream()), "gt_gcn_layer_bwd")
        lin_w, lin_b, root, bn_w, bn_b = params[:5]
        edge_params = params[5:]
        D = desc.D
        off = 0

        def take(n, shape):
            nonlocal off
            t = grads[off:off + n].view(shape)
            off += n
            return t

        g_lin_w, g_lin_b, g_root = take(D * D, lin_w.shape), take(D, lin_b.shape), take(D, root.shape)
        g_edge = []
        if ctx.spec.kind == "linear":
            g_edge = [take(edge_params[0].numel(), edge_params[0].shape), take(D, edge_params[1].shape)]
        elif ctx.spec.kind == "tables":
            for t in edge_params:
                g_edge.append(take(t.numel(), t.shape))
        g_bn_w, g_bn_b = take(D, bn_w.shape), take(D, bn_b.shape)
        return (d_h, d_vn, None, None, None, None, None, None, None, None, g_lin_w, g_lin_b, g_root, g_bn_w, g_bn_b, *g_edge)


def gcn_layer_eligible(conv, bn, h, spec, drop_ratio, training):
    from .modules.conv import GCNConv
    return (isinstance(conv, GCNConv) and h.is_cuda and h.dtype == torch.float32 and h.shape[1] % 4 == 0
            and spec.kind in ("linear", "tables", "none")
            and bn.affine and bn.track_running_stats and bn.momentum is not None and not _any_sync(bn))


def gcn_layer(h_in, vn, gs, conv, bn, spec, relu, residual, training, dropout_p=0.0, seed=0):
    """-> (x, y); x is h_in when vn is None."""
    if spec.kind == "linear":
        edge_params = [spec.weight, spec.bias]
    elif spec.kind == "tables":
        edge_params = list(spec.table_list)
    else:
        edge_params = []
    x, y = _GcnLayer.apply(h_in, vn, gs, spec, relu, residual, training, bn, dropout_p, seed, conv.linear.weight, conv.linear.bias,
                           conv.root_emb.weight, bn.weight, bn.bias, *edge_params)
    return (x if vn is not None else h_in), y


# ------------------------------------------------------------------------------------------------
# virtual-node update
# ------------------------------------------------------------------------------------------------
class _VnUpdate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, vn, gs, residual, training, bn1, bn2, dropout_p, seed, *params):
        L = _bind()
        x, vn = x.contiguous(), vn.contiguous()
        N, D = x.shape
        desc = VnUpdateDesc()
        desc.N, desc.B, desc.D = N, gs.B, D
        desc.residual, desc.training, desc.compute = int(residual), int(training), _compute_code()
        desc.bn_momentum, desc.bn_eps = float(bn1.momentum), float(bn1.eps)
        desc.dropout_p, desc.seed = (float(dropout_p) if training else 0.0), int(seed)
        ident = getattr(gs, "_identity", None)
        desc.graph_ptr, desc.node_graph = _ptr(gs.graph_ptr), _ptr(gs.node_graph)
        if residual:
            if ident is None:
                ident = torch.arange(gs.B, dtype=torch.int32, device=x.device)
            desc.identity_graph = _ptr(ident)
        for name, p in zip(("w1", "b1", "bn1_w", "bn1_b", "w2", "b2", "bn2_w", "bn2_b"), params):
            setattr(desc, name, _ptr(_f32c(p)))
        desc.bn1_rm, desc.bn1_rv, desc.bn1_nbt = _ptr(bn1.running_mean), _ptr(bn1.running_var), _ptr(bn1.num_batches_tracked)
        desc.bn2_rm, desc.bn2_rv, desc.bn2_nbt = _ptr(bn2.running_mean), _ptr(bn2.running_var), _ptr(bn2.num_batches_tracked)
        dev = x.device
        saved = _bytes(L.gt_vn_update_saved_bytes(C.byref(desc)), dev)
        ws_bytes = L.gt_vn_update_workspace_bytes(C.byref(desc))
        ws = _bytes(ws_bytes, dev)
        out = torch.empty_like(vn)
        _lib.check(L.gt_vn_update_fwd(C.byref(desc), _ptr(x), _ptr(vn), _ptr(out), _ptr(saved), _ptr(ws), ws_bytes, _stream()),
                   "gt_vn_update_fwd")
        ctx.save_for_backward(saved, *params)
        ctx.desc, ctx.ident, ctx.xshape = desc, ident, x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        L = _bind()
        saved = ctx.saved_tensors[0]
        params = ctx.saved_tensors[1:]
        desc = ctx.desc
        dev = saved.device
        g = g.contiguous()
        d_x = torch.empty(ctx.xshape, dtype=torch.float32, device=dev)
        d_vn = torch.empty((desc.B, desc.D), dtype=torch.float32, device=dev)
        grads = torch.empty(L.gt_vn_update_grad_elems(C.byref(desc)), dtype=torch.float32, device=dev)
        ws_bytes = L.gt_vn_update_workspace_bytes(C.byref(desc))
        ws = _bytes(ws_bytes, dev)
        _lib.check(L.gt_vn_update_bwd(C.byref(desc), _ptr(g), _ptr(saved), None, _ptr(d_x), _ptr(d_vn), _ptr(grads), _ptr(ws),
                                      ws_bytes, _stream()), "gt_vn_update_bwd")
        return (d_x, d_vn, None, None, None, None, None, None, None, *_split(grads, params))


def vn_update_eligible(seq, x, drop_ratio, training):
    from .modules.norm import BatchNorm1d
    mods = list(seq)
    return (len(mods) == 6 and isinstance(mods[0], torch.nn.Linear) and isinstance(mods[1], BatchNorm1d)
            and isinstance(mods[3], torch.nn.Linear) and isinstance(mods[4], BatchNorm1d) and x.is_cuda
            and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and not _any_sync(mods[1], mods[4]))


def vn_update(x, vn, gs, seq, residual, training, dropout_p=0.0, seed=0):
    m = list(seq)
    return _VnUpdate.apply(x, vn, gs, residual, training, m[1], m[4], dropout_p, seed, m[0].weight, m[0].bias, m[1].weight, m[1].bias,
                           m[3].weight, m[3].bias, m[4].weight, m[4].bias)
