"""Fused model path: the whole GNNTransformer forward (and its backward) as ONE autograd node and ONE C call per direction.

The module-by-module path (modules/, layers.py) costs ~3.5 ms of host time per Code2 step in Python autograd bookkeeping alone;
rounds 1-3 sequenced the library's layer composites from here (~60 ctypes calls, descriptor refresh, arena book-keeping: ~2.3 ms
of interpreter time per step).  Now the sequencing lives behind `gt_model_forward` / `gt_model_backward`
(csrc/model.hip, include/graphtrans_hip.h "Whole-model driver"): this module fills one `gt_model` struct per model (static
pointers, sizes, gradient offsets, streams, events) and one `gt_model_batch` per step, allocates the two arenas the driver asks for,
and hands the flat gradient buffer's slices to the parameters.  Autograd sees a single node.

Covered configuration (everything else keeps using the module path, see `eligible`):
  GNN_node / GNN_node_Virtualnode with GCNConv or GINConv layers, Linear(<=4, D), BondEncoder-style
  embedding tables or "zero" edge encoders,
  any gnn_dropout, JK in {last, cat}, ASTNodeEncoder / AtomEncoder / nn.Linear inputs, no perturb;
  packed token layout (cls / last pooling, no positional encoder, no masked layers), ReLU / GELU post-norm
  encoder layers; stacked max_seq_len heads or a single head.
Reference call path: trainers/base_trainer.py:29-36 -> models/gnn_transformer.py:88-127 -> modules/gnn_module.py:181-224 ->
modules/transformer_encoder.py:42-61.
"""
import ctypes as C
import os
import sys
import threading
import weakref

import numpy as np
import torch

from . import _lib, layers
from ._lib import GT_BF16, GT_EDGE_LINEAR, GT_EDGE_NONE, GT_EDGE_TABLES, GT_F32
from .graph import _stream


OVERLAP_VN = os.environ.get("GT_OVERLAP_VN", "1") != "0"
OVERLAP_DW = os.environ.get("GT_OVERLAP_DW", "1") != "0"
# gt_graph_prep (8 short launches) and the encoder's weight images do not depend on anything the step computes: they run on a side
# stream beside the zero-fills, the bf16x3 images, the input embedding and layer 0's GEMM (csrc/model.hip)
PREP_OVERLAP = os.environ.get("GT_PREP_OVERLAP", "1") != "0"
# the overlap stream pays for itself only when the kernels are long enough to hide its extra stream operations
DW_OVERLAP_MIN_ELEMS = 1 << 20   # nodes x emb_dim of the batch
VN_DEFER_DW = os.environ.get("GT_VN_DEFER_DW", "1") != "0"

MAXL, MAXT = 16, 16   # GT_MODEL_MAX_LAYERS, GT_MODEL_MAX_TABLES
_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


def _c4(n):
    return (n + 3) // 4 * 4


# ---- mirrors of the driver's structs (include/graphtrans_hip.h; sizes checked against gt_model_abi_sizes at first use) ----------
class ImageSet(C.Structure):   # gt_image_set
    _fields_ = [("n_jobs", _i32), ("n_bind", _i32)] + [(k, _vp) for k in ("job_w", "job_N", "job_K", "job_T", "job_img", "bind_w",
                                                                          "bind_N", "bind_K", "bind_f", "bind_t")]


class StageRingDesc(C.Structure):   # gt_stage_ring
    _fields_ = [("base", _vp), ("slot_bytes", _i64), ("slots", _i32), ("next", _i32), ("events", _vp * 64)]


class ModelDesc(C.Structure):   # gt_model
    _fields_ = [(k, _i32) for k in ("conv", "L", "n_enc", "has_vn", "jk_cat", "residual", "embed_kind", "n_tables", "vn0_in_embed",
                                    "embed_sorted", "with_cls", "vn_defer_dw")] + \
               [(k, _i64) for k in ("D", "d", "Nh", "ldy", "ne_K", "max_input_len", "dw_overlap_min_elems")] + \
               [("conv_layers", _vp), ("vn", _vp), ("enc", _vp), ("tables", _vp * MAXT), ("table_rows", _i64 * MAXT),
                ("table_clamp", _i64 * MAXT)] + \
               [(k, _vp) for k in ("vn_emb", "ne_w", "ne_b", "g2t_w", "g2t_b", "cls", "nin_w", "nin_b", "nout_w", "nout_b", "head_w",
                                   "head_b", "zero_i64")] + \
               [("nin_eps", _f32), ("nout_eps", _f32), ("off_tables", _i64 * MAXT), ("off_ne_w", _i64), ("off_ne_b", _i64),
                ("off_vn_emb", _i64), ("off_conv", _i64 * MAXL), ("off_vn", _i64 * MAXL), ("off_g2t_w", _i64), ("off_g2t_b", _i64),
                ("off_cls", _i64), ("off_nin_w", _i64), ("off_nin_b", _i64), ("off_enc", _i64 * MAXL), ("off_nout_w", _i64),
                ("off_nout_b", _i64), ("off_head_w", _i64), ("off_head_b", _i64), ("grad_total", _i64),
                ("st_vn", _vp), ("st_dw", _vp), ("st_prep", _vp), ("ev_x", _vp * MAXL), ("ev_vn", _vp * MAXL), ("ev_dvn", _vp * MAXL),
                ("ev_extra", _vp * MAXL), ("ev_pool", _vp * MAXL), ("ev_vnemb", _vp), ("ev_sort", _vp * 2), ("ev_wt", _vp * 2),
                ("ev_prep_begin", _vp), ("ev_graph", _vp), ("ev_w1", _vp), ("w3", ImageSet), ("w3_enc", ImageSet), ("w1", ImageSet),
                ("pna_src", _vp), ("pna_img", _vp), ("pna_map", _vp), ("pna_inv", _vp), ("pna_n_img", _i64), ("pna_n_src", _i64),
                ("off_pna_src", _i64), ("pna_img_off", (_i64 * 4) * MAXL), ("pna_kinds", _i32 * 8), ("pna_avg_log", _f32),
                ("pna_avg_lin", _f32)]


class BatchDesc(C.Structure):   # gt_model_batch
    _fields_ = [("N", _i64), ("E", _i64), ("B", _i64), ("edge_index", _vp), ("batch", _vp), ("sizes_host", _vp)] + \
               [(k, _vp) for k in ("graph_ptr", "node_graph", "in_ptr", "in_src", "in_eid", "out_ptr", "out_dst", "out_eid", "deg", "dis",
                                   "seq_desc", "last_rows", "work_items")] + \
               [("rows", _i64), ("max_npos", _i64), ("num_work", _i64), ("lay_exact", _i32), ("pad0_", _i32),
                ("x", _vp), ("x_stride0", _i64), ("x_stride1", _i64), ("node_depth", _vp), ("depth_stride", _i64), ("edge_attr", _vp),
                ("zeros_B", _vp), ("ident_B", _vp), ("ptr01", _vp)] + \
               [(k, _i32) for k in ("training", "compute", "tdt", "will_bwd", "use_w3", "use_w1", "sync_bn", "pad2_")] + \
               [("gnn_p", _f32), ("enc_p", _f32), ("gnn_seed", C.c_uint64), ("enc_seed", C.c_uint64), ("ring", _vp)]


class SizesDesc(C.Structure):   # gt_model_sizes
    _fields_ = [(k, _i64) for k in ("rows", "max_npos", "num_work", "arena_bytes", "barena_bytes")] + [("exact", _i32), ("pad_", _i32)]


_ABI_OK = []


def _check_abi():
    if not _ABI_OK:
        out = (_i64 * 4)()
        _lib.check(_lib.lib().gt_model_abi_sizes(out), "gt_model_abi_sizes")
        mine = (C.sizeof(ModelDesc), C.sizeof(BatchDesc), C.sizeof(ImageSet), C.sizeof(StageRingDesc))
        if tuple(out) != mine:
            raise RuntimeError("graphtrans_amd.engine: struct layouts differ from the library's (%r vs %r): rebuild the library" % (mine, tuple(out)))
        _ABI_OK.append(True)


# The overlap streams are created ONCE per device and shared by every plan: HIP maps streams onto a handful of hardware queues in
# creation order, and the streams of a second model's plan landed on the main stream's queue (measured: the second model built
# in a process ran 8 % slower, whichever precision mode it used).
_SIDE_STREAMS = {}
# priorities of the side streams: the virtual-node chain is short and latency-bound and sits on the critical path of the
# backward (highest); the weight-gradient GEMMs are long, chip-filling and nobody waits for them (lowest)
_SIDE_LEVEL = {"vn": int(os.environ.get("GT_PRIO_VN", "-1")), "dw": int(os.environ.get("GT_PRIO_DW", "1"))}


def _side_stream(device, which):
    """a HIP stream created by the library (gt_stream_create: carries a priority); lives as long as the process"""
    key = (torch.device(device).index or 0, which)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        with torch.cuda.device(device):
            st = _lib.lib().gt_stream_create(_SIDE_LEVEL.get(which, 0))
        if not st:
            raise RuntimeError("gt_stream_create failed")
        _SIDE_STREAMS[key] = st
    return st


# ---- pinned staging ring for the host-built token layout (one per device; the driver takes a slot per forward) -------------------
_RINGS = {}
_RING_LOCK = threading.Lock()
RING_SLOTS, RING_SLOT_BYTES = 64, 1 << 17


def _ring(device):
    key = torch.device(device).index or 0
    r = _RINGS.get(key)
    if r is None:
        lib = _lib.lib()
        buf = torch.empty(RING_SLOTS * RING_SLOT_BYTES, dtype=torch.uint8).pin_memory()   # pinning costs ~1 ms: once
        d = StageRingDesc()
        d.base, d.slot_bytes, d.slots, d.next = buf.data_ptr(), RING_SLOT_BYTES, RING_SLOTS, 0
        with torch.cuda.device(device):
            for i in range(RING_SLOTS):
                d.events[i] = lib.gt_event_create()
        r = _RINGS[key] = (d, buf)
    return r[0]


def _arr(ctype, vals):
    return (ctype * max(len(vals), 1))(*vals)


def _image_set(imgs):
    """gt_image_set view of a w3.W3Images / W1Images object (its ctypes arrays stay owned by `imgs`)"""
    s = ImageSet()
    if imgs is None:
        return s
    s.n_jobs, s.n_bind = imgs._n, len(imgs.weights)
    for k, a in (("job_w", imgs._w), ("job_N", imgs._N), ("job_K", imgs._K), ("job_T", imgs._T), ("job_img", imgs._img),
                 ("bind_w", imgs._bw), ("bind_N", imgs._bN), ("bind_K", imgs._bK), ("bind_f", imgs._bf), ("bind_t", imgs._bt)):
        setattr(s, k, C.cast(a, _vp))
    return s


PNA_TOWER_IMAGES = os.environ.get("GT_PNA_TOWER_IMAGES", "1") != "0"   # bf16x3 images of the PNA tower weights (grouped k_lin3)
W_MAX_BOUND = 64   # W3_MAX_BOUND / W1_MAX_BOUND of the library's bind tables


# ---------------------------------------------------------------------------------------------------
# plan: parameter order, gradient layout, the driver's model struct (built once per model)
# ---------------------------------------------------------------------------------------------------
class _Plan:
    def __init__(self, model):
        from .modules.conv import GINConv
        from .modules.gnn_module import GNN_node_Virtualnode
        from . import w3
        _check_abi()
        lib = _lib.lib()
        gnn, enc = model.gnn_node, model.transformer_encoder
        self.L, self.has_vn = gnn.num_layer, isinstance(gnn, GNN_node_Virtualnode)
        if hasattr(gnn, "layers"):   # PNANodeEmbedding
            self.kind = "pna"
            self.D = gnn.layers[0].in_channels
            self.jk_cat = False
        else:
            self.kind = "gin" if isinstance(gnn.convs[0], GINConv) else "gcn"
            self.D = gnn.convs[0].emb_dim
            self.jk_cat = gnn.JK == "cat"
        self.d = enc.d_model
        self.dev = next(model.parameters()).device
        self.total = 0
        self.params = []   # (param, offset)
        self._cache = {}   # per batch size: small index arrays
        self._keep = []    # ctypes arrays / tensors the struct points into
        L, D, d = self.L, self.D, self.d

        def seg(p):
            off = self.total
            self.params.append((p, off))
            self.total += _c4(p.numel())
            return off

        cm = self.cm = ModelDesc()
        ne = gnn.node_encoder
        if hasattr(ne, "type_encoder"):  # ASTNodeEncoder
            self.embed = [ne.type_encoder.weight, ne.attribute_encoder.weight, ne.depth_encoder.weight]
            self.embed_clamp = [-1, -1, int(ne.max_depth)]
            self.embed_kind = "ast"
        elif type(ne) is torch.nn.Linear:   # TU datasets / the ER stress: dense float features (dataset/tud.py:65)
            self.embed, self.embed_clamp, self.embed_kind = [], [], "linear"
            self.ne_lin = ne
            self.ne_K = int(ne.in_features)
        else:
            self.embed = [e.weight for e in ne.atom_embedding_list]
            self.embed_clamp = [-1] * len(self.embed)
            self.embed_kind = "atom"
        cm.embed_kind = {"atom": 0, "linear": 1, "ast": 2}[self.embed_kind]
        cm.n_tables = len(self.embed)
        for t, w in enumerate(self.embed):
            cm.off_tables[t] = seg(w)
        cm.off_ne_w = cm.off_ne_b = -1
        if self.embed_kind == "linear":
            cm.off_ne_w, cm.off_ne_b = seg(ne.weight), seg(ne.bias)
            cm.ne_K = self.ne_K
        self.vn_emb = gnn.virtualnode_embedding.weight if self.has_vn else None
        cm.off_vn_emb = seg(self.vn_emb) if self.has_vn else -1
        self.w3_weights = []
        self.etab_flat = None
        if self.kind == "pna":
            self._init_pna(gnn, seg)
        else:
            self._init_convs(gnn, seg)
        nvn = L - 1 if self.has_vn else 0
        self.vn_desc = (layers.VnUpdateDesc * max(nvn, 1))()
        if self.has_vn:
            for l, seq in enumerate(gnn.mlp_virtualnode_list):
                m = list(seq)
                off = None
                ps = [m[0].weight, m[0].bias, m[1].weight, m[1].bias, m[3].weight, m[3].bias, m[4].weight, m[4].bias]
                for p in ps:
                    o = seg(p)
                    off = o if off is None else off
                cm.off_vn[l] = off
                desc = self.vn_desc[l]
                desc.D = D
                for name, p in zip(("w1", "b1", "bn1_w", "bn1_b", "w2", "b2", "bn2_w", "bn2_b"), ps):
                    setattr(desc, name, p.data_ptr())
                desc.bn1_rm, desc.bn1_rv, desc.bn1_nbt = m[1].running_mean.data_ptr(), m[1].running_var.data_ptr(), m[1].num_batches_tracked.data_ptr()
                desc.bn2_rm, desc.bn2_rv, desc.bn2_nbt = m[4].running_mean.data_ptr(), m[4].running_var.data_ptr(), m[4].num_batches_tracked.data_ptr()
                desc.bn_momentum, desc.bn_eps = float(m[1].momentum), float(m[1].eps)
        g2t = self.g2t = model.gnn2transformer
        cm.off_g2t_w, cm.off_g2t_b = seg(g2t.weight), seg(g2t.bias)
        self.w3_weights.append(g2t.weight)
        self.cls = enc.cls_embedding
        cm.off_cls = seg(self.cls) if self.cls is not None else -1
        self.norm_in = enc.norm_input
        cm.off_nin_w, cm.off_nin_b = (seg(enc.norm_input.weight), seg(enc.norm_input.bias)) if enc.norm_input is not None else (-1, -1)
        self.enc_layers = list(enc.transformer.layers)
        self.enc_desc = (layers.EncoderLayerDesc * max(len(self.enc_layers), 1))()
        self.w3_enc_weights = []   # the encoder layers' GEMMs run in fp32 only in the fp32 mode (fp32 token rows)
        for i, mod in enumerate(self.enc_layers):
            off = None
            ps = layers.encoder_layer_params(mod)
            for p in ps:
                o = seg(p)
                off = o if off is None else off
            cm.off_enc[i] = off
            desc = self.enc_desc[i]
            desc.d_model, desc.ffn, desc.nhead = d, mod.linear1.weight.shape[0], enc.nhead
            for name, p in zip(("in_w", "in_b", "out_w", "out_b", "l1_w", "l1_b", "l2_w", "l2_b", "n1_w", "n1_b", "n2_w", "n2_b"), ps):
                setattr(desc, name, p.data_ptr())
            desc.ln_eps = float(mod.norm1.eps)
            desc.act = layers.ENC_ACT[enc.activation]
            self.w3_enc_weights += [mod.self_attn.in_proj_weight, mod.self_attn.out_proj.weight, mod.linear1.weight, mod.linear2.weight]
        self.norm_out = enc.transformer.norm
        cm.off_nout_w, cm.off_nout_b = (seg(self.norm_out.weight), seg(self.norm_out.bias)) if self.norm_out is not None else (-1, -1)
        if model.max_seq_len is None:
            self.heads = [model.graph_pred_linear]
        else:
            self.heads = list(model.graph_pred_linear_list)
        self.num_tasks = model.num_tasks
        self.Nh = sum(h.weight.shape[0] for h in self.heads)
        self.ldy = _c4(self.Nh)
        # head gradients: one [Nh][d] block and one [Nh] block; the per-head grads are slices of them
        cm.off_head_w = self.total
        for h in self.heads:
            self.params.append((h.weight, self.total))
            self.total += h.weight.numel()
        self.total = _c4(self.total)
        cm.off_head_b = self.total
        for h in self.heads:
            self.params.append((h.bias, self.total))
            self.total += h.bias.numel()
        self.total = _c4(self.total)
        # the max_seq_len prediction heads (models/gnn_transformer.py:120-126) run as ONE GEMM over their stacked weights: the
        # parameters' storage IS the stacked matrix -- each head's weight / bias becomes a view of it
        self.head_w_flat, self.head_b_flat = self.heads[0].weight, self.heads[0].bias
        if len(self.heads) > 1:
            with torch.no_grad():
                wf = torch.cat([h.weight.detach() for h in self.heads]).contiguous()
                bf = torch.cat([h.bias.detach() for h in self.heads]).contiguous()
                r0 = 0
                for h in self.heads:
                    n_ = h.weight.shape[0]
                    h.weight.data = wf[r0:r0 + n_]
                    h.bias.data = bf[r0:r0 + n_]
                    r0 += n_
            self.head_w_flat, self.head_b_flat = wf, bf
        cm.grad_total = self.total
        # persistent flat gradient buffer and its per-parameter views
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=self.dev)
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for p, o in self.params]
        self.plist = [p for p, _ in self.params]
        self.param_ptrs = tuple(p.data_ptr() for p in self.plist)
        # ---- the rest of the static struct
        cm.conv = {"gcn": 0, "gin": 1, "pna": 2}[self.kind]
        cm.L, cm.n_enc, cm.has_vn, cm.jk_cat, cm.residual = L, len(self.enc_layers), int(self.has_vn), int(self.jk_cat), int(bool(gnn.residual))
        cm.D, cm.d, cm.Nh, cm.ldy = D, d, self.Nh, self.ldy
        cm.max_input_len = int(enc.max_input_len)
        cm.with_cls = int(self.cls is not None)
        cm.vn_defer_dw = int(VN_DEFER_DW)
        cm.conv_layers = C.cast(self.conv_desc, _vp)
        cm.vn = C.cast(self.vn_desc, _vp) if nvn else None
        cm.enc = C.cast(self.enc_desc, _vp)
        for t, w in enumerate(self.embed):
            cm.tables[t], cm.table_rows[t], cm.table_clamp[t] = w.data_ptr(), int(w.shape[0]), self.embed_clamp[t]
        # x_0 = h_0 + vn_0[batch] with vn_0 = the ONE row of virtualnode_embedding for every graph (gnn_module.py:195):
        # the embedding-sum kernel takes it as one more table whose index column is a stride-0 zero
        self.vn0_in_embed = self.has_vn and self.embed_kind != "linear" and len(self.embed) < 16
        cm.vn0_in_embed = int(self.vn0_in_embed)
        self.embed_sorted = bool(self.embed) and max(int(t.shape[0]) for t in self.embed) <= 16384
        cm.embed_sorted = int(self.embed_sorted)
        self.zero_i64 = torch.zeros(1, dtype=torch.int64, device=self.dev)
        cm.zero_i64 = self.zero_i64.data_ptr()
        cm.vn_emb = self.vn_emb.data_ptr() if self.has_vn else None
        if self.embed_kind == "linear":
            cm.ne_w, cm.ne_b = ne.weight.data_ptr(), ne.bias.data_ptr()
        cm.g2t_w, cm.g2t_b = g2t.weight.data_ptr(), g2t.bias.data_ptr()
        cm.cls = self.cls.data_ptr() if self.cls is not None else None
        if self.norm_in is not None:
            cm.nin_w, cm.nin_b, cm.nin_eps = self.norm_in.weight.data_ptr(), self.norm_in.bias.data_ptr(), float(self.norm_in.eps)
        if self.norm_out is not None:
            cm.nout_w, cm.nout_b, cm.nout_eps = self.norm_out.weight.data_ptr(), self.norm_out.bias.data_ptr(), float(self.norm_out.eps)
        cm.head_w, cm.head_b = self.head_w_flat.data_ptr(), self.head_b_flat.data_ptr()
        # streams: the virtual-node update of layer l only feeds layer l+1 (second stream beside layer l's conv / BatchNorm +
        # aggregate backward); the weight-gradient GEMMs run on a third stream beside the dX chain; the graph structure on a fourth
        self.side = _side_stream(self.dev, "vn") if (self.has_vn and OVERLAP_VN) else None
        self.side_dw = _side_stream(self.dev, "dw") if OVERLAP_DW else None
        self.side_prep = _side_stream(self.dev, "prep") if PREP_OVERLAP else None
        cm.st_vn, cm.st_dw, cm.st_prep = self.side, self.side_dw, self.side_prep
        self._events = []

        def ev():
            e = lib.gt_event_create()
            self._events.append(e)
            return e
        with torch.cuda.device(self.dev):
            for l in range(L):
                cm.ev_x[l], cm.ev_vn[l], cm.ev_dvn[l], cm.ev_extra[l], cm.ev_pool[l] = ev(), ev(), ev(), ev(), ev()
            cm.ev_vnemb, cm.ev_prep_begin, cm.ev_graph, cm.ev_w1 = ev(), ev(), ev(), ev()
            cm.ev_sort[0], cm.ev_sort[1], cm.ev_wt[0], cm.ev_wt[1] = ev(), ev(), ev(), ev()
        self._set_min_elems()
        # ---- weight images: bf16x3 (fp32-accurate big-M GEMMs on the bf16 pipe, csrc/linear3x.h) and fragment-order bf16 (encoder
        # layers, csrc/linear1.h); rebuilt by the driver at every forward (the optimizer changed the weights: one launch each), bound
        # per host thread inside the driver.  A weight list beyond the bind tables' 64 entries keeps the exact / tiled kernels.
        if self.embed_kind == "linear" and _c4(self.ne_K) == self.ne_K:
            self.w3_weights.append(self.ne_lin.weight)
        self.imgs3 = self.imgs3e = self.imgs1 = None
        if w3.ENABLED:
            if len(self.w3_weights) <= W_MAX_BOUND:
                self.imgs3 = w3.W3Images(self.w3_weights)
            if self.w3_enc_weights and len(self.w3_weights) + len(self.w3_enc_weights) <= W_MAX_BOUND:
                self.imgs3e = w3.W3Images(self.w3_weights + self.w3_enc_weights)
        if w3.W1_ENABLED and self.w3_enc_weights and len(self.w3_enc_weights) <= W_MAX_BOUND:
            self.imgs1 = w3.W1Images(self.w3_enc_weights)
        cm.w3, cm.w3_enc, cm.w1 = _image_set(self.imgs3), _image_set(self.imgs3e), _image_set(self.imgs1)
        self.ctx_bytes = int(lib.gt_model_ctx_bytes())
        lo, hi = (_i64 * 3)(), (_i64 * 3)()
        _lib.check(lib.gt_model_grad_ranges(C.byref(cm), lo, hi), "gt_model_grad_ranges")
        self.ranges = list(zip(lo, hi))
        self.cm_ref = C.byref(cm)

    def _init_convs(self, gnn, seg):
        from .modules.conv import GINConv
        cm, L, D = self.cm, self.L, self.D
        # ---- conv layers.  Gradient block order of gt_gcn_layer_bwd: lin_w, lin_b, root, edge_w, edge_b, bn_w, bn_b;
        # of gt_gin_layer_bwd: eps (20-float slot), edge tables | edge_w, edge_b, w1, b1, bn1_w, bn1_b, w2, b2, bn_w, bn_b
        self.convs = list(zip(gnn.convs, gnn.batch_norms))
        # BondEncoder-style tables: the aggregate kernels read ONE [rows][D] matrix per layer.  Instead of stacking a layer's tables
        # by a torch.cat per step, the parameters' storage IS the stacked matrix -- each table's weight becomes a view of it (same
        # Parameter objects, same state_dict keys; optim.FusedAdamW notices the move and rebuilds its tables)
        self.etab_flat = None
        tabs_all = [[t.weight for t in getattr(conv.edge_encoder, "bond_embedding_list", [])] for conv, _ in self.convs]
        if any(tabs_all):
            with torch.no_grad():
                flat = torch.cat([t.detach() for tl in tabs_all for t in tl]).contiguous()
                r0 = 0
                for tl in tabs_all:
                    for t in tl:
                        n_ = int(t.shape[0])
                        t.data = flat[r0:r0 + n_]
                        r0 += n_
            self.etab_flat = flat
        self.conv_desc = ((layers.GinLayerDesc if self.kind == "gin" else layers.GcnLayerDesc) * L)()
        for l, (conv, bn) in enumerate(self.convs):
            desc = self.conv_desc[l]
            ee = conv.edge_encoder
            tabs = tabs_all[l]
            if tabs:
                edge, mode = tabs, "tables"
            elif isinstance(ee, torch.nn.Module):
                edge, mode = [ee.weight, ee.bias], "linear"
            else:
                edge, mode = [], None
            if self.kind == "gcn":
                plist = [conv.linear.weight, conv.linear.bias, conv.root_emb.weight, *edge, bn.weight, bn.bias]
                self.w3_weights.append(conv.linear.weight)
            else:
                m = list(conv.mlp)
                plist = [conv.eps, *edge, m[0].weight, m[0].bias, m[1].weight, m[1].bias, m[3].weight, m[3].bias, bn.weight, bn.bias]
                self.w3_weights += [m[0].weight, m[3].weight]
            off = None
            for p in plist:
                o = seg(p)
                off = o if off is None else off
                if self.kind == "gin" and p is conv.eps:
                    self.total = o + 20   # d_eps + the aggregate backward's scratch (GIN_EPS_SLOT in layers.hip)
            cm.off_conv[l] = off
            desc.D = D
            if self.kind == "gcn":
                desc.lin_w, desc.lin_b, desc.root = conv.linear.weight.data_ptr(), conv.linear.bias.data_ptr(), conv.root_emb.weight.data_ptr()
            else:
                desc.eps = conv.eps.data_ptr()
                desc.w1, desc.b1, desc.bn1_w, desc.bn1_b = m[0].weight.data_ptr(), m[0].bias.data_ptr(), m[1].weight.data_ptr(), m[1].bias.data_ptr()
                desc.w2, desc.b2 = m[3].weight.data_ptr(), m[3].bias.data_ptr()
                desc.bn1_rm, desc.bn1_rv, desc.bn1_nbt = m[1].running_mean.data_ptr(), m[1].running_var.data_ptr(), m[1].num_batches_tracked.data_ptr()
            if mode == "linear":
                desc.edge_mode = GT_EDGE_LINEAR
                desc.edge_cols = ee.weight.shape[1]
                desc.edge_w, desc.edge_b = ee.weight.data_ptr(), ee.bias.data_ptr()
            elif mode == "tables":
                desc.edge_mode = GT_EDGE_TABLES
                desc.edge_cols, desc.table_rows = len(tabs), sum(int(t.shape[0]) for t in tabs)
                acc = 0
                for i, t in enumerate(tabs):
                    desc.tab_off[i] = acc
                    acc += int(t.shape[0])
                desc.edge_w = tabs[0].data_ptr()   # the layer's stacked [rows][D] block
            else:
                desc.edge_mode = GT_EDGE_NONE
            desc.bn_w, desc.bn_b = bn.weight.data_ptr(), bn.bias.data_ptr()
            desc.bn_rm, desc.bn_rv, desc.bn_nbt = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()
            desc.bn_momentum, desc.bn_eps = float(bn.momentum), float(bn.eps)

    _SCALER_KIND = {None: 0, "identity": 0, "amplification": 1, "attenuation": 2, "linear": 3, "inverse_linear": 4}

    def _init_pna(self, gnn, seg):
        """PNANodeEmbedding (modules/pna/pna_module.py:16-78): per layer lin / BatchNorm gradients in the layer block of the flat
        buffer; the tower weights of ALL layers live in one flat storage (the parameters become views of it), from which the driver
        rebuilds the re-stacked images [A ; B], [b | 0], the per-scaler post-Linear blocks and their bias with ONE gather per step."""
        from .modules.pna.pna_module import _AGG_SLOT
        cm, L, D = self.cm, self.L, self.D
        convs = list(gnn.layers)
        bns = [b.module for b in gnn.batch_norms]
        self.convs = list(zip(convs, bns))
        c0 = convs[0]
        T, F = c0.towers, c0.F_in
        A, S = len(c0.aggregators), len(c0._blocks)
        first = S - len(c0.scalers)
        cols = (A * len(c0.scalers) + 1) * F
        self.conv_desc = (layers.PnaLayerDesc * L)()
        # ---- flat tower storage: per layer [pre_w (T,F,2F) | pre_b (T,F) | post_w (T,F,cols) | post_b (T,F)]
        per_layer = T * F * 2 * F + T * F + T * F * cols + T * F
        n_src = L * per_layer
        with torch.no_grad():
            flat = torch.empty(n_src, dtype=torch.float32, device=self.dev)
            off = 0
            tower_params = []   # (param, offset inside the flat storage)
            for conv in convs:
                for group, shape in ((lambda m: m[0].weight, (F, 2 * F)), (lambda m: m[0].bias, (F,))):
                    for t in range(T):
                        p = group(conv.pre_nns[t])
                        n_ = p.numel()
                        flat[off:off + n_].copy_(p.detach().reshape(-1))
                        p.data = flat[off:off + n_].view(shape)
                        tower_params.append((p, off))
                        off += n_
                for group, shape in ((lambda m: m[0].weight, (F, cols)), (lambda m: m[0].bias, (F,))):
                    for t in range(T):
                        p = group(conv.post_nns[t])
                        n_ = p.numel()
                        flat[off:off + n_].copy_(p.detach().reshape(-1))
                        p.data = flat[off:off + n_].view(shape)
                        tower_params.append((p, off))
                        off += n_
            assert off == n_src
        self.pna_flat = flat
        # ---- image layout per layer: [pre_stack (T,2F,F) | pre_b2 (T,2F) | Wst (T,S*F,5F) | bst (T,S*F)] and the gather map
        img_per_layer = T * 2 * F * F + T * 2 * F + T * S * F * 5 * F + T * S * F
        n_img = L * img_per_layer
        mp = np.full(n_img, -1, dtype=np.int64)
        t_ = np.arange(T).reshape(T, 1, 1)
        for l in range(L):
            sb = l * per_layer                      # source bases of this layer
            s_pre_w, s_pre_b = sb, sb + T * F * 2 * F
            s_post_w, s_post_b = s_pre_b + T * F, s_pre_b + T * F + T * F * cols
            ib = l * img_per_layer
            i_pre_w, i_pre_b = ib, ib + T * 2 * F * F
            i_post_w, i_post_b = i_pre_b + T * 2 * F, i_pre_b + T * 2 * F + T * S * F * 5 * F
            r = np.arange(2 * F).reshape(1, 2 * F, 1)
            k = np.arange(F).reshape(1, 1, F)
            src = s_pre_w + t_ * (F * 2 * F) + np.where(r < F, r, r - F) * (2 * F) + np.where(r < F, 0, F) + k
            mp[i_pre_w:i_pre_w + T * 2 * F * F] = src.reshape(-1)
            rb = np.arange(2 * F).reshape(1, 2 * F)
            srcb = np.where(rb < F, s_pre_b + np.arange(T).reshape(T, 1) * F + rb, -1)
            mp[i_pre_b:i_pre_b + T * 2 * F] = srcb.reshape(-1)
            # post: block s of tower t, output o, operand column c in [x | mean | max | min | std] (modules/pna/pna_module.py:_post_maps)
            w = np.full((T, S * F, 5 * F), -1, dtype=np.int64)
            o = np.arange(F).reshape(F, 1)
            f = np.arange(F).reshape(1, F)
            for t in range(T):
                tb = s_post_w + t * F * cols
                for s_ in range(S):
                    rows = slice(s_ * F, (s_ + 1) * F)
                    if s_ == 0:
                        w[t, rows, 0:F] = tb + o * cols + f
                    if s_ < first:
                        continue
                    for ai, a in enumerate(c0.aggregators):
                        slot = _AGG_SLOT[a]
                        w[t, rows, F + slot * F:F + (slot + 1) * F] = tb + o * cols + F + (s_ - first) * A * F + ai * F + f
            mp[i_post_w:i_post_w + T * S * F * 5 * F] = w.reshape(-1)
            bb = np.full((T, S * F), -1, dtype=np.int64)
            bb[:, :F] = s_post_b + np.arange(T).reshape(T, 1) * F + np.arange(F).reshape(1, F)
            mp[i_post_b:i_post_b + T * S * F] = bb.reshape(-1)
            for j, v in enumerate((i_pre_w, i_pre_b, i_post_w, i_post_b)):
                cm.pna_img_off[l][j] = v
        inv = np.full(n_src, -1, dtype=np.int64)
        used = mp >= 0
        inv[mp[used]] = np.nonzero(used)[0]
        assert (inv >= 0).all(), "every tower weight sits in the images exactly once"
        self.pna_map = torch.from_numpy(mp.astype(np.int32)).to(self.dev)
        self.pna_inv = torch.from_numpy(inv.astype(np.int32)).to(self.dev)
        self.pna_img = torch.zeros(n_img, dtype=torch.float32, device=self.dev)
        # ---- gradient layout: per layer [lin_w, lin_b, bn_w, bn_b] (gt_pna_layer_bwd's order), then the flat tower storage
        for l, (conv, bn) in enumerate(self.convs):
            off0 = None
            for p in (conv.lin.weight, conv.lin.bias, bn.weight, bn.bias):
                o_ = seg(p)
                off0 = o_ if off0 is None else off0
            cm.off_conv[l] = off0
            self.w3_weights.append(conv.lin.weight)
            desc = self.conv_desc[l]
            desc.D, desc.T, desc.S = D, T, S
            ib = l * img_per_layer
            base = self.pna_img.data_ptr()
            desc.pre_w, desc.pre_b = base + 4 * cm.pna_img_off[l][0], base + 4 * cm.pna_img_off[l][1]
            desc.post_w, desc.post_b = base + 4 * cm.pna_img_off[l][2], base + 4 * cm.pna_img_off[l][3]
            desc.lin_w, desc.lin_b = conv.lin.weight.data_ptr(), conv.lin.bias.data_ptr()
            desc.bn_w, desc.bn_b = bn.weight.data_ptr(), bn.bias.data_ptr()
            desc.bn_rm, desc.bn_rv, desc.bn_nbt = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()
            desc.bn_momentum, desc.bn_eps = float(bn.momentum), float(bn.eps)
        # the towers' re-stacked matrices get bf16x3 images too (grouped k_lin3: the T GEMMs of a tower stack in one launch on the
        # bf16 matrix pipe at fp32 accuracy); views of pna_img, whose contents the driver gathers before it builds the images
        self.pna_tower_views = []
        if PNA_TOWER_IMAGES:
            for l in range(L):
                for j, (rows_, cols_) in ((0, (2 * F, F)), (2, (S * F, 5 * F))):
                    o0 = int(cm.pna_img_off[l][j])
                    for t in range(T):
                        v = self.pna_img[o0 + t * rows_ * cols_:o0 + (t + 1) * rows_ * cols_].view(rows_, cols_)
                        self.pna_tower_views.append(v)
            # the bind table holds W_MAX_BOUND weights: when the tower views would overflow it they are the ones left out (the towers
            # then run the exact-fp32 tiled kernels) -- NOT every image of the model (the whole set silently fell back)
            if len(self.w3_weights) + len(self.pna_tower_views) + 2 <= W_MAX_BOUND:   # (+ gnn2transformer, + a Linear node encoder)
                self.w3_weights.extend(self.pna_tower_views)
            else:
                import warnings
                warnings.warn("graphtrans_amd: %d PNA tower matrices do not fit the library's %d-entry weight-image table beside the "
                              "model's other weights: the towers run on the exact-fp32 kernels" % (len(self.pna_tower_views), W_MAX_BOUND))
                self.pna_tower_views = []
        cm.off_pna_src = self.total
        for p, o_ in tower_params:
            self.params.append((p, self.total + o_))
        self.total += n_src
        cm.pna_src, cm.pna_img = flat.data_ptr(), self.pna_img.data_ptr()
        cm.pna_map, cm.pna_inv = self.pna_map.data_ptr(), self.pna_inv.data_ptr()
        cm.pna_n_img, cm.pna_n_src = n_img, n_src
        for i, blk in enumerate(c0._blocks):
            cm.pna_kinds[i] = self._SCALER_KIND[blk]
        cm.pna_avg_log, cm.pna_avg_lin = float(c0.avg_deg["log"]), float(c0.avg_deg["lin"])

    def _set_min_elems(self):
        self.min_elems = DW_OVERLAP_MIN_ELEMS
        self.cm.dw_overlap_min_elems = DW_OVERLAP_MIN_ELEMS

    def __del__(self):
        try:
            lib = _lib.lib()
            for e in self._events:
                lib.gt_event_destroy(e)
        except Exception:
            pass

    def small(self, B):
        c = self._cache.get(B)
        if c is None:
            z = torch.zeros(B, dtype=torch.int32, device=self.dev)
            i = torch.arange(B, dtype=torch.int32, device=self.dev)
            p = torch.tensor([0, B], dtype=torch.int32, device=self.dev)
            c = self._cache[B] = (z, i, p, z.data_ptr(), i.data_ptr(), p.data_ptr())
        return c


# Per-model engine state (plan with its ctypes descriptors / HIP events / streams, eligibility cache, attached
# GradSync) lives OUTSIDE the module: nothing unpicklable ends up in `model.__dict__`, so copy.deepcopy(model),
# torch.save(model) and EMA snapshots keep working after the first fused forward, and a copy builds its own plan.
_STATE = weakref.WeakKeyDictionary()


def state(model):
    st = _STATE.get(model)
    if st is None:
        st = _STATE[model] = {}
    return st


def invalidate(model):
    """Forget the cached plan / eligibility of `model` (parameters were frozen, replaced or re-registered)."""
    st = _STATE.get(model)
    if st is not None:
        st.pop("plan", None)
        st.pop("eligible", None)
        st.pop("params", None)
        st.pop("bn_sync", None)
        st.pop("bn_hook", None)


def _bn_sync_hook(model, plan):
    """dist.BnSyncHook of a model whose BatchNorms are synchronised (modules/norm.py:convert_sync_batchnorm), else None"""
    from .modules.norm import BatchNorm1d, any_sync
    st = state(model)
    if "bn_sync" not in st:
        bns = [m for m in model.modules() if isinstance(m, BatchNorm1d)]
        st["bn_sync"] = bns[0].sync_group if (bns and any_sync(*bns)) else False
    grp = st["bn_sync"]
    if grp is False:
        return None
    hook = st.get("bn_hook")
    if hook is None:
        from .dist import BnSyncHook
        hook = st["bn_hook"] = BnSyncHook(grp)
    return hook


def _plan(model):
    st = state(model)
    plan = st.get("plan")
    if plan is None or plan.param_ptrs != tuple(p.data_ptr() for p in plan.plist):
        plan = st["plan"] = _Plan(model)
    return plan


def eligible(model, batched_data, perturb):
    """True when the fused path covers this model / call (the static part is cached per model and mode)."""
    if perturb is not None or not getattr(model, "fused", True):
        return False
    # the fused node differentiates EVERY parameter and assigns `.grad` itself: any frozen parameter (epoch_callback's freeze_gnn,
    # or a user's requires_grad_(False) on any submodule), any tensor hook on a parameter and a DistributedDataParallel wrapper send
    # the model through the module path.  All three are looked at on EVERY call (a model may be wrapped or hooked after its first
    # fused forward): flags and hooks by one pass over the parameters, the wrapper through a flag that torch's module-registration
    # hook sets at the moment a DistributedDataParallel takes the model as its `.module` (`_note_ddp_wrapper`).
    st = state(model)
    plist = st.get("params")
    if plist is None:
        plist = st["params"] = list(model.parameters())
    flags = []
    for p in plist:
        if p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
            return False
        flags.append(p.requires_grad)
    w = st.get("ddp_wrapper")   # set by the module-registration hook below when a DistributedDataParallel adopts this model
    if w is not None:
        if w() is not None:
            return False
        st["ddp_wrapper"] = None   # the wrapper is gone
    key = (model.training, torch.is_grad_enabled(), tuple(flags))
    cache = st.setdefault("eligible", {})
    ok = cache.get(key)
    if ok is None:
        ok = cache[key] = _eligible_static(model)
    if not ok:
        return False
    x = batched_data.x
    gnn = model.gnn_node
    ne = gnn.node_encoder
    if type(ne) is torch.nn.Linear:
        # (features that require a gradient go through the module path: the fused node only differentiates parameters)
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == ne.in_features
                and not x.requires_grad and getattr(batched_data, "node_depth", None) is None):
            return False
    else:
        if not (x.is_cuda and x.dtype == torch.int64 and x.dim() == 2):
            return False
        if hasattr(ne, "type_encoder"):   # ASTNodeEncoder reads columns 0, 1 of x and node_depth (dataset/utils.py:28-30)
            nd = getattr(batched_data, "node_depth", None)
            if x.shape[1] < 2 or nd is None or nd.numel() != x.shape[0]:
                return False
        elif x.shape[1] != len(ne.atom_embedding_list):
            # e.g. the reference's `--feature simple` (dataset/mol.py:65-69) slices x to 2 columns: the fused kernels
            # index one column per table, so a different column count goes through the module path
            return False
    if hasattr(gnn, "layers"):   # PNANodeEmbedding: no edge features on its path (modules/pna/pna_module.py:73)
        return True
    # edge features: the aggregate kernels read `edge_cols` values per edge at that pitch
    ee = gnn.convs[0].edge_encoder
    ea = getattr(batched_data, "edge_attr", None)
    tabs = getattr(ee, "bond_embedding_list", None)
    if tabs is not None:
        if ea is None or ea.dim() != 2 or ea.shape[1] != len(tabs) or ea.dtype != torch.int64 or not ea.is_cuda:
            return False
    elif isinstance(ee, torch.nn.Linear):
        if ea is None or ea.dim() != 2 or ea.shape[1] != ee.in_features or not ea.is_cuda or not ea.is_floating_point():
            return False
    return True


def has_grad_hooks(model):
    """Tensor hooks on parameters (register_hook / register_post_accumulate_grad_hook): the fused node assigns `.grad` itself and
    would never fire them."""
    for p in model.parameters():
        if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
            return True
    return False


def wrapped_in_ddp(model):
    """True when `model` is the `.module` of a live torch DistributedDataParallel instance.  DDP reduces gradients from hooks on the
    parameters' AccumulateGrad nodes (C++ side, invisible from here), which a node that assigns `.grad` directly never reaches:
    such a model runs the module-by-module path (autograd accumulates as usual), or -- the supported way -- uses
    graphtrans_amd.dist.GradSync, whose all-reduce the fused backward issues itself."""
    w = state(model).get("ddp_wrapper")
    return w is not None and w() is not None


def _note_ddp_wrapper(parent, name, child):
    """torch.nn module-registration hook (fires on `parent.<name> = child`, i.e. inside DistributedDataParallel.__init__): remember
    the wrapper -- weakly -- in the wrapped model's state.  Exact and free on the step (the previous detection compared
    sys.getrefcount(model) between calls, which any extra reference -- a list, a closure, a debugger -- triggered or hid)."""
    if name == "module" and isinstance(child, torch.nn.Module):
        from torch.nn.parallel import DistributedDataParallel
        if isinstance(parent, DistributedDataParallel):
            state(child)["ddp_wrapper"] = weakref.ref(parent)
    return None


torch.nn.modules.module.register_module_module_registration_hook(_note_ddp_wrapper)


def _eligible_static(model):
    from . import ops
    from .modules.conv import GCNConv
    from .modules.norm import BatchNorm1d
    gnn, enc = model.gnn_node, model.transformer_encoder
    if hasattr(gnn, "layers"):
        return _eligible_static_pna(model)
    try:
        if not model._use_packed() or gnn.JK not in ("last", "cat"):
            return False
        if gnn.num_layer > MAXL or len(enc.transformer.layers) > MAXL or len(enc.transformer.layers) < 1:   # the driver's fixed-size tables
            return False
        ne = gnn.node_encoder
        if type(ne) is torch.nn.Linear:
            if ne.bias is None:
                return False
        elif not (hasattr(ne, "type_encoder") or hasattr(ne, "atom_embedding_list")):
            return False
        if hasattr(ne, "atom_embedding_list") and len(ne.atom_embedding_list) > 16:
            return False
        D = gnn.convs[0].emb_dim
        if D % 4:
            return False
        from .modules.conv import GINConv, tables_fit_lds
        kinds = {type(conv) for conv in gnn.convs}
        if len(kinds) != 1 or not (kinds <= {GCNConv, GINConv}):
            return False
        for conv, bn in zip(gnn.convs, gnn.batch_norms):
            if not isinstance(bn, BatchNorm1d):
                return False
            if not (bn.affine and bn.track_running_stats and bn.momentum is not None):
                return False
            if isinstance(conv, GINConv):
                m = list(conv.mlp)
                if not (len(m) == 4 and isinstance(m[0], torch.nn.Linear) and isinstance(m[1], BatchNorm1d)
                        and isinstance(m[3], torch.nn.Linear) and m[1].affine and m[1].track_running_stats
                        and m[1].momentum is not None):
                    return False
            ee = conv.edge_encoder
            tabs = getattr(ee, "bond_embedding_list", None)
            if tabs is not None:
                rows = sum(int(t.weight.shape[0]) for t in tabs)
                if len(tabs) > 4 or not tables_fit_lds(rows, D):
                    return False
            elif isinstance(ee, torch.nn.Module):
                if not (isinstance(ee, torch.nn.Linear) and ee.in_features <= 4 and ee.bias is not None):
                    return False
            else:
                e = ee(None)
                if not (isinstance(e, (int, float)) and e == 0):
                    return False
        if hasattr(gnn, "mlp_virtualnode_list"):
            for seq in gnn.mlp_virtualnode_list:
                m = list(seq)
                if not (len(m) == 6 and isinstance(m[0], torch.nn.Linear) and isinstance(m[1], BatchNorm1d)
                        and isinstance(m[3], torch.nn.Linear) and isinstance(m[4], BatchNorm1d)):
                    return False
        if enc.activation not in layers.ENC_ACT or enc.d_model % 8 or enc.compute_dtype not in (torch.float32, torch.bfloat16):
            return False
        for mod in enc.transformer.layers:
            if mod.linear1.weight.shape[0] % 8:
                return False
        if model.gnn2transformer.weight.shape[1] % 4:
            return False
        # (synchronised BatchNorm -- statistics over all data-parallel ranks -- runs on this path too: the library's BatchNorm calls
        # exchange their statistics through dist.BnSyncHook, installed around the pass; all of the model's BatchNorms or none)
        from .modules.norm import any_sync
        bns = [m for m in model.modules() if isinstance(m, BatchNorm1d)]
        if any_sync(*bns) and not all(getattr(b, "sync", False) for b in bns):
            return False
        if any_sync(*bns) and len({id(getattr(b, "sync_group", None)) for b in bns}) != 1:
            return False
        for p in model.parameters():
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.requires_grad):
                return False
    except Exception:
        return False
    return True


def _eligible_static_pna(model):
    """PNATransformer (models/pna_transformer.py:16-118) on the fused path: PNANodeEmbedding with the residual connection, towers
    that divide the input, single-layer pre / post nets, packed token layout."""
    from .modules.norm import BatchNorm1d, any_sync
    from .modules.pna.pna_module import PNAConv, _AGG_SLOT
    gnn, enc = model.gnn_node, model.transformer_encoder
    try:
        if not (model.pooling in ("cls", "last") and getattr(model, "layout", "auto") != "padded" and len(enc.transformer.layers) > 0):
            return False
        convs = list(gnn.layers)
        if not convs or len(convs) > MAXL or len(enc.transformer.layers) > MAXL or not gnn.residual:
            return False
        c0 = convs[0]
        for c in convs:
            if not (isinstance(c, PNAConv) and c.divide_input and c.in_channels == c.out_channels == c0.in_channels and c.towers == c0.towers
                    and c.F_in == c.F_out and c.aggregators == c0.aggregators and c.scalers == c0.scalers and c.avg_deg == c0.avg_deg):
                return False
        if len(c0._blocks) > 8 or any(a not in _AGG_SLOT for a in c0.aggregators) or c0.in_channels > 1024:
            return False
        if any(b not in _Plan._SCALER_KIND for b in c0._blocks):
            return False
        for bn in gnn.batch_norms:
            m = getattr(bn, "module", None)
            if not (isinstance(m, BatchNorm1d) and m.affine and m.track_running_stats and m.momentum is not None):
                return False
        ne = gnn.node_encoder
        if type(ne) is torch.nn.Linear:
            if ne.bias is None:
                return False
        elif not (hasattr(ne, "type_encoder") or hasattr(ne, "atom_embedding_list")):
            return False
        if hasattr(ne, "atom_embedding_list") and len(ne.atom_embedding_list) > 16:
            return False
        if model.gnn2transformer.in_features != c0.in_channels or c0.in_channels % 4 or c0.F_in % 4:
            return False
        if enc.activation not in layers.ENC_ACT or enc.d_model % 8 or enc.compute_dtype not in (torch.float32, torch.bfloat16):
            return False
        for mod in enc.transformer.layers:
            if mod.linear1.weight.shape[0] % 8:
                return False
        bns = [m for m in model.modules() if isinstance(m, BatchNorm1d)]
        if any_sync(*bns) and not all(getattr(b, "sync", False) for b in bns):
            return False
        if any_sync(*bns) and len({id(getattr(b, "sync_group", None)) for b in bns}) != 1:
            return False
        for p in model.parameters():
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.requires_grad):
                return False
    except Exception:
        return False
    return True


# ---------------------------------------------------------------------------------------------------
# the autograd node
# ---------------------------------------------------------------------------------------------------
def _call(name, *args):
    _lib.check(getattr(_lib.lib(), name)(*args), name)


def _num_graphs(batched_data, sizes):
    ng = getattr(batched_data, "_num_graphs", None)
    if ng is None and sizes is None and hasattr(type(batched_data), "num_graphs"):
        try:
            ng = batched_data.num_graphs
        except Exception:
            ng = None
    if ng is None:
        ng = len(sizes) if sizes is not None else int(batched_data.batch[-1].item()) + 1   # (device sync: the reference's own, gnn_module.py:195)
    return int(ng)


class _FusedModel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trigger, model, batched_data, plan):
        from . import ops
        lib = _lib.lib()
        cm = plan.cm
        if plan.min_elems != DW_OVERLAP_MIN_ELEMS:
            plan._set_min_elems()
        enc, gnn = model.transformer_encoder, model.gnn_node
        training = bool(model.training)
        bt = BatchDesc()
        x, batch = batched_data.x, batched_data.batch
        N = int(batch.numel())
        keep = [x, batch]
        gs = getattr(batched_data, "_gt_structure", None)   # built earlier by the module path (modules/gnn_module.py:batch_structure): reused
        sizes = getattr(batched_data, "_sizes", None)
        if gs is not None:
            if gs.ready_event is not None:
                _call("gt_stream_wait_event", _stream(), gs.ready_event)
                gs.ready_event = None
            E, B = gs.E, gs.B
            for k in ("graph_ptr", "node_graph", "in_ptr", "in_src", "in_eid", "out_ptr", "out_dst", "out_eid", "deg", "dis"):
                setattr(bt, k, getattr(gs, k).data_ptr())
            keep.append(gs)
            if sizes is None:
                sizes = gs._sizes
            lay = gs._layouts.get(("packed", int(enc.max_input_len), enc.cls_embedding is not None))
            if lay is not None:
                bt.seq_desc, bt.last_rows = lay.desc.data_ptr(), lay.last_rows.data_ptr()
                work = getattr(lay, "work", None)
                bt.work_items = work.data_ptr() if work is not None else None
                bt.rows, bt.max_npos, bt.num_work, bt.lay_exact = lay.rows, lay.max_npos, getattr(lay, "num_work", 0), int(lay.exact)
                keep.append(lay)
        else:
            ei = batched_data.edge_index
            if ei.dtype != torch.int64 or batch.dtype != torch.int64:
                raise TypeError("edge_index and batch must be int64 (PyG collation dtype)")
            ei, batch = ei.contiguous(), batch.contiguous()
            E, B = int(ei.shape[1]), _num_graphs(batched_data, sizes)
            bt.edge_index, bt.batch = ei.data_ptr(), batch.data_ptr()
            keep += [ei, batch]
        if sizes is not None:
            sizes = np.ascontiguousarray(sizes, dtype=np.int64)
            if sizes.size != B:
                raise ValueError("per-graph sizes do not match the number of graphs")
            bt.sizes_host = sizes.ctypes.data
        bt.N, bt.E, bt.B = N, E, B
        # ---- inputs
        if plan.embed_kind == "linear":
            x = x.contiguous()
            bt.x = x.data_ptr()
            keep.append(x)
        else:
            bt.x, bt.x_stride0, bt.x_stride1 = x.data_ptr(), x.stride(0), x.stride(1)
            if plan.embed_kind == "ast":
                depth = batched_data.node_depth.reshape(-1)
                bt.node_depth, bt.depth_stride = depth.data_ptr(), (depth.stride(0) if N > 1 else 1)
                keep.append(depth)
        ea = getattr(batched_data, "edge_attr", None)
        mode0 = plan.conv_desc[0].edge_mode if plan.kind != "pna" else GT_EDGE_NONE
        if mode0 == GT_EDGE_LINEAR:
            ea = ea if (ea.dtype == torch.float32 and ea.is_contiguous()) else ea.float().contiguous()
        elif mode0 == GT_EDGE_TABLES:
            if ea.dtype != torch.int64:
                raise TypeError("embedding-table edge encoders need int64 edge_attr")
            ea = ea.contiguous()
        if mode0 != GT_EDGE_NONE:
            bt.edge_attr = ea.data_ptr()
            keep.append(ea)
        if plan.has_vn:
            sm = plan.small(B)
            bt.zeros_B, bt.ident_B, bt.ptr01 = sm[3], sm[4], sm[5]
        # compute type of the fp32-stored GEMMs (message passing, gnn2transformer, heads): exact-fp32 MFMA or bf16 MFMA
        # (ops.set_matmul_dtype); the encoder layers compute in bf16 whenever their token rows are stored in bf16
        # (layers.hip: dtype == bf16 ? bf16 : compute), i.e. (fp32, bf16 tokens) is the mixed mode of bench.py
        bf16_mm = ops.get_matmul_dtype() == torch.bfloat16
        tok_bf16 = enc.compute_dtype == torch.bfloat16
        bt.training, bt.compute, bt.tdt = int(training), (GT_BF16 if bf16_mm else GT_F32), (GT_BF16 if tok_bf16 else GT_F32)
        will_bwd = bool(ctx.needs_input_grad[0])
        bt.will_bwd = int(will_bwd)
        # exact-fp32 GEMM mode: the big-M linears run as bf16x6 on the bf16 matrix pipe on images of their weights; bf16 token rows:
        # the encoder layers' GEMMs run with the weight stationary in registers on fragment-order images
        if not bf16_mm and N >= 1024:
            if tok_bf16:
                bt.use_w3 = 1 if plan.imgs3 is not None else 0
            else:
                bt.use_w3 = 2 if plan.imgs3e is not None else (1 if plan.imgs3 is not None else 0)
        bt.use_w1 = int(tok_bf16 and plan.imgs1 is not None)
        # dropout seeds: drawn in the module path's order (GNN first, then the encoder)
        from .modules.gnn_module import _gnn_seed
        bt.gnn_seed = _gnn_seed(gnn) & 0xFFFFFFFFFFFFFFFF
        bt.gnn_p = float(gnn.drop_ratio) if training else 0.0
        bt.enc_p = float(enc.dropout_p) if training else 0.0
        bt.enc_seed = (int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF) if (training and enc.dropout_p > 0) else 0
        hook = _bn_sync_hook(model, plan) if training else None
        bt.sync_bn = int(hook is not None)
        ring = _ring(plan.dev)
        bt.ring = C.addressof(ring)
        cbuf = C.create_string_buffer(plan.ctx_bytes)
        sz = SizesDesc()
        with _RING_LOCK:
            _lib.check(lib.gt_model_prepare(plan.cm_ref, C.byref(bt), cbuf, C.byref(sz)), "gt_model_prepare")
        arena = (torch.empty if sz.exact else torch.zeros)(sz.arena_bytes, dtype=torch.uint8, device=plan.dev)
        logits = torch.empty((B, plan.ldy), dtype=torch.float32, device=plan.dev)
        if hook is not None:
            hook.install()
        try:
            _lib.check(lib.gt_model_forward(plan.cm_ref, cbuf, arena.data_ptr(), logits.data_ptr(), _stream()), "gt_model_forward")
        finally:
            if hook is not None:
                hook.uninstall()
                if hook.error is not None:
                    e, hook.error = hook.error, None
                    raise e
        if will_bwd:
            ctx.state = (plan, cbuf, arena, int(sz.barena_bytes), bool(sz.exact), keep, state(model).get("sync"), hook, B)
        else:
            ctx.state = None
        ctx.set_materialize_grads(False)
        return logits[:, :plan.Nh] if plan.ldy != plan.Nh else logits

    @staticmethod
    def backward(ctx, dlogits):
        s = ctx.state
        if s is None:
            raise RuntimeError("graphtrans_amd fused model: backward through the graph a second time "
                               "(the saved activations are freed after the first backward)")
        if dlogits is None:
            return None, None, None, None
        plan, cbuf, arena, barena_bytes, exact, _keep, model_sync, hook, B = s
        from . import ops
        lib = _lib.lib()
        dl = dlogits.reshape(B, plan.Nh)
        if dl.dtype != torch.float32:
            dl = dl.to(torch.float32)
        dl = ops._padded_rows(dl, plan.ldy) if plan.ldy != plan.Nh else dl.contiguous()
        # gradients: straight into the persistent flat buffer when nothing has to be accumulated
        plist = plan.plist
        direct = True
        for p in plist:
            if p.grad is not None:
                direct = False
                break
        flat = plan.flat if direct else torch.empty_like(plan.flat)
        barena = (torch.empty if exact else torch.zeros)(barena_bytes, dtype=torch.uint8, device=plan.dev)
        st = _stream()
        args = (plan.cm_ref, cbuf, dl.data_ptr(), flat.data_ptr(), barena.data_ptr())
        sync = model_sync if (direct and model_sync is not None and model_sync.active) else None
        if hook is not None:
            hook.install()
        try:
            if sync is None:
                _lib.check(lib.gt_model_backward(*args, 7, st), "gt_model_backward")
            else:
                # data parallel: each stage completes one range of the flat buffer, which goes on the wire (asynchronously, on RCCL's
                # stream) while the next stage runs: heads .. gnn2transformer | message passing | input encoder
                # (the C side keeps its overlap and deferred-reduce sections open between the stage calls on this host thread: a raise
                # from Python in between -- the collective, the BatchNorm hook -- must not leave them pointing at this step's arena)
                staged_open = False
                try:
                    for i, stage in enumerate((1, 2, 4)):
                        staged_open = stage != 4
                        _lib.check(lib.gt_model_backward(*args, stage, st), "gt_model_backward")
                        if stage == 2 and plan.has_vn and plan.side is not None:
                            _call("gt_stream_wait_event", st, plan.cm.ev_vnemb)   # d virtualnode_embedding was reduced on the second stream
                        if stage != 4:
                            _call("gt_overlap_dw_sync")   # (the last stage joins the weight-gradient stream itself)
                        lo, hi = plan.ranges[i]
                        sync.reduce_flat(flat, lo, hi)
                    staged_open = False
                finally:
                    if staged_open:
                        lib.gt_defer_begin(None, 0)
                        lib.gt_overlap_dw_end()
        finally:
            if hook is not None:
                hook.uninstall()
                if hook.error is not None:
                    e, hook.error = hook.error, None
                    raise e
        # ---- hand the gradients to the parameters
        if direct:
            for p, v in zip(plist, plan.views):
                p.grad = v
        else:
            for (p, o_), v0 in zip(plan.params, plan.views):
                v = flat[o_:o_ + p.numel()].view(p.shape)
                p.grad = v if p.grad is None else p.grad + v
        ctx.state = None
        return None, None, None, None


def forward(model, batched_data):
    """logits (B, Nh) [row-padded storage] of the fused path; `model` must be `eligible`."""
    plan = _plan(model)
    return _FusedModel.apply(plan.plist[0], model, batched_data, plan)
