"""Fused model path: the whole GNNTransformer forward (and its backward) as ONE autograd node.

The module-by-module path (modules/, layers.py) costs ~3.5 ms of host time per Code2 step in Python
autograd bookkeeping alone (tools/host_phases.py): ~80 autograd Functions, each saving a dozen
parameter tensors and returning a dozen gradients.  Here the same C-ABI entry points are called back
to back on buffers carved out of one arena, parameter gradients are written straight into one flat
buffer whose slices become `p.grad`, and autograd sees a single node.

Covered configuration (everything else keeps using the module path, see `eligible`):
  GNN_node / GNN_node_Virtualnode with GCNConv or GINConv layers, Linear(<=4, D), BondEncoder-style
  embedding tables or "zero" edge encoders,
  any gnn_dropout, JK in {last, cat}, ASTNodeEncoder / AtomEncoder inputs, no perturb;
  packed token layout (cls / last pooling, no positional encoder, no masked layers), ReLU post-norm
  encoder layers; stacked max_seq_len heads or a single head.
Reference call path: models/gnn_transformer.py:88-127 -> modules/gnn_module.py:181-224 ->
modules/transformer_encoder.py:42-61.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib, layers
from ._lib import GT_BF16, GT_EDGE_LINEAR, GT_EDGE_NONE, GT_EDGE_TABLES, GT_F32
from .graph import _stream


OVERLAP_VN = os.environ.get("GT_OVERLAP_VN", "1") != "0"
OVERLAP_DW = os.environ.get("GT_OVERLAP_DW", "1") != "0"


def _c4(n):
    return (n + 3) // 4 * 4


class _Bump:
    """Byte offsets inside one arena (256-byte aligned)."""

    def __init__(self):
        self.off = 0

    def take(self, nbytes):
        o = self.off
        self.off = (o + int(nbytes) + 255) // 256 * 256
        return o


# The two overlap streams are created ONCE per device and shared by every plan: HIP maps streams onto a handful of
# hardware queues in creation order, and the streams of a second model's plan landed on the main stream's queue
# (measured: the second model built in a process ran 8 % slower, whichever precision mode it used).
_SIDE_STREAMS = {}


class _Stream:
    """a HIP stream created by the library (gt_stream_create: carries a priority); lives as long as the process"""

    def __init__(self, device, level):
        with torch.cuda.device(device):
            self.cuda_stream = _lib.lib().gt_stream_create(level)
        if not self.cuda_stream:
            raise RuntimeError("gt_stream_create failed")


# priorities of the two side streams: the virtual-node chain is short and latency-bound and sits on the critical path of
# the backward (highest); the weight-gradient GEMMs are long, chip-filling and nobody waits for them (lowest)
DW_OVERLAP_MIN_ELEMS = 1 << 20   # nodes x emb_dim of the batch
VN_DEFER_DW = os.environ.get("GT_VN_DEFER_DW", "1") != "0"
_SIDE_LEVEL = {"vn": int(os.environ.get("GT_PRIO_VN", "-1")), "dw": int(os.environ.get("GT_PRIO_DW", "1"))}


def _side_stream(device, which):
    key = (torch.device(device).index or 0, which)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = _Stream(device, _SIDE_LEVEL.get(which, 0))
    return st


# ---- step preamble beside the first kernels -------------------------------------------------------------------------------
# gt_graph_prep (8 short launches: degree count, scans, fill, per-node sorts) and the encoder's weight images do not depend on
# anything the step computes, and nothing needs them before the first aggregate / the first encoder layer: they run on a side
# stream that starts where the main stream stands when the batch arrives (so memory the allocator hands out for them is no
# longer in use) while the main stream does its zero-fills, the bf16x3 images, the input embedding and layer 0's GEMM.
PREP_OVERLAP = os.environ.get("GT_PREP_OVERLAP", "1") != "0"
_PREPS = {}


class Prep:
    def __init__(self, device):
        lib = _lib.lib()
        self.device = torch.device(device)
        self.stream = _side_stream(device, "prep").cuda_stream
        self.ev_begin, self.ev_graph, self.ev_w1 = lib.gt_event_create(), lib.gt_event_create(), lib.gt_event_create()
        self.active = False

    def begin(self):
        """the side stream continues from the main stream's current position"""
        main = _stream()
        _call("gt_event_record", self.ev_begin, main)
        _call("gt_stream_wait_event", self.stream, self.ev_begin)
        self.active = True

    def graph_done(self):
        _call("gt_event_record", self.ev_graph, self.stream)
        return self.ev_graph

    def w1_done(self):
        _call("gt_event_record", self.ev_w1, self.stream)
        self.active = False
        return self.ev_w1


def prep_for(device):
    if not PREP_OVERLAP or torch.device(device).type != "cuda":
        return None
    key = torch.device(device).index or 0
    p = _PREPS.get(key)
    if p is None:
        p = _PREPS[key] = Prep(device)
    return p


# ---------------------------------------------------------------------------------------------------
# plan: parameter order, gradient layout, persistent descriptors (built once per model)
# ---------------------------------------------------------------------------------------------------
class _Plan:
    def __init__(self, model):
        from .modules.gnn_module import GNN_node_Virtualnode
        gnn, enc = model.gnn_node, model.transformer_encoder
        self.L, self.has_vn = gnn.num_layer, isinstance(gnn, GNN_node_Virtualnode)
        self.D = gnn.convs[0].emb_dim
        self.d = enc.d_model
        self.jk_cat = gnn.JK == "cat"
        self.dev = next(model.parameters()).device
        self.total = 0
        self.params = []   # (param, offset)
        self._cache = {}   # per batch size: small index arrays

        def seg(p):
            off = self.total
            self.params.append((p, off))
            self.total += _c4(p.numel())
            return off

        ne = gnn.node_encoder
        if hasattr(ne, "type_encoder"):  # ASTNodeEncoder
            self.embed = [ne.type_encoder.weight, ne.attribute_encoder.weight, ne.depth_encoder.weight]
            self.embed_clamp = [-1, -1, int(ne.max_depth)]
            self.embed_kind = "ast"
        elif type(ne) is torch.nn.Linear:   # TU datasets / the ER stress: dense float features (dataset/tud.py:65)
            self.embed, self.embed_clamp, self.embed_kind = [], [], "linear"
            self.ne_lin = ne
            self.ne_K = int(ne.in_features)
            self.ne_Kp = _c4(self.ne_K)
        else:
            self.embed = [e.weight for e in ne.atom_embedding_list]
            self.embed_clamp = [-1] * len(self.embed)
            self.embed_kind = "atom"
        self.embed_off = [seg(t) for t in self.embed]
        if self.embed_kind == "linear":
            self.ne_off = [seg(ne.weight), seg(ne.bias)]
        self.vn_emb = gnn.virtualnode_embedding.weight if self.has_vn else None
        self.vn_emb_off = seg(self.vn_emb) if self.has_vn else None
        # conv layers.  Gradient block order of gt_gcn_layer_bwd: lin_w, lin_b, root, edge_w, edge_b, bn_w, bn_b;
        # of gt_gin_layer_bwd: eps (20-float slot), edge tables | edge_w, edge_b, w1, b1, bn1_w, bn1_b, w2, b2, bn_w, bn_b
        from .modules.conv import GINConv
        self.kind = "gin" if isinstance(gnn.convs[0], GINConv) else "gcn"
        self.gcn, self.gcn_off, self.gcn_edge = [], [], []   # gcn_edge: "linear" | "tables" | None
        self.tables, self.tab_off, self.table_rows = [], [], []
        for conv, bn in zip(gnn.convs, gnn.batch_norms):
            ee = conv.edge_encoder
            tabs = getattr(ee, "bond_embedding_list", None)
            if tabs is not None:
                edge, mode = [t.weight for t in tabs], "tables"
                offs, acc = [], 0
                for t in tabs:
                    offs.append(acc)
                    acc += int(t.weight.shape[0])
                self.tab_off.append(offs)
                self.table_rows.append(acc)
            elif isinstance(ee, torch.nn.Module):
                edge, mode = [ee.weight, ee.bias], "linear"
                self.tab_off.append([])
                self.table_rows.append(0)
            else:
                edge, mode = [], None
                self.tab_off.append([])
                self.table_rows.append(0)
            self.tables.append(edge if mode == "tables" else [])
            self.gcn_edge.append(mode)
            self.gcn.append((conv, bn))
            if self.kind == "gcn":
                plist = [conv.linear.weight, conv.linear.bias, conv.root_emb.weight, *edge, bn.weight, bn.bias]
            else:
                m = list(conv.mlp)
                plist = [conv.eps, *edge, m[0].weight, m[0].bias, m[1].weight, m[1].bias, m[3].weight, m[3].bias, bn.weight, bn.bias]
            off = None
            for p in plist:
                o = seg(p)
                off = o if off is None else off
                if self.kind == "gin" and p is conv.eps:
                    self.total = o + 20   # d_eps + the aggregate backward's scratch (GIN_EPS_SLOT in layers.hip)
            self.gcn_off.append(off)
        self.vn, self.vn_off = [], []
        if self.has_vn:
            for seq in gnn.mlp_virtualnode_list:
                m = list(seq)
                off = None
                for p in [m[0].weight, m[0].bias, m[1].weight, m[1].bias, m[3].weight, m[3].bias, m[4].weight, m[4].bias]:
                    o = seg(p)
                    off = o if off is None else off
                self.vn.append(m)
                self.vn_off.append(off)
        g2t = model.gnn2transformer
        self.g2t = g2t
        self.g2t_off = (seg(g2t.weight), seg(g2t.bias))
        self.cls = enc.cls_embedding
        self.cls_off = seg(self.cls) if self.cls is not None else None
        self.norm_in = enc.norm_input
        self.norm_in_off = (seg(enc.norm_input.weight), seg(enc.norm_input.bias)) if enc.norm_input is not None else None
        self.enc_layers, self.enc_off = list(enc.transformer.layers), []
        self.enc_act = layers.ENC_ACT[enc.activation]
        for mod in self.enc_layers:
            off = None
            for p in layers.encoder_layer_params(mod):
                o = seg(p)
                off = o if off is None else off
            self.enc_off.append(off)
        self.norm_out = enc.transformer.norm
        self.norm_out_off = (seg(self.norm_out.weight), seg(self.norm_out.bias)) if self.norm_out is not None else None
        if model.max_seq_len is None:
            self.heads = [model.graph_pred_linear]
        else:
            self.heads = list(model.graph_pred_linear_list)
        self.num_tasks = model.num_tasks
        self.Nh = sum(h.weight.shape[0] for h in self.heads)
        self.ldy = _c4(self.Nh)
        # head gradients: one [Nh][d] block and one [Nh] block; the per-head grads are slices of them
        self.headw_off = self.total
        for h in self.heads:
            self.params.append((h.weight, self.total))
            self.total += h.weight.numel()
        self.total = _c4(self.total)
        self.headb_off = self.total
        for h in self.heads:
            self.params.append((h.bias, self.total))
            self.total += h.bias.numel()
        self.total = _c4(self.total)
        # the max_seq_len prediction heads (models/gnn_transformer.py:120-126) run as ONE GEMM over their stacked weights: instead
        # of stacking them by a 12.8 MB torch.cat per step (two launches on the critical path), the parameters' storage IS the
        # stacked matrix -- each head's weight / bias becomes a view of it (same Parameter objects, same state_dict keys; an
        # optimizer that caches data pointers, like optim.FusedAdamW, notices the move and rebuilds its tables)
        self.head_w_flat = self.head_b_flat = None
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
        # persistent flat gradient buffer and its per-parameter views
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=self.dev)
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for p, o in self.params]
        self.plist = [p for p, _ in self.params]
        self.param_ptrs = tuple(p.data_ptr() for p in self.plist)
        # persistent descriptors (batch-dependent fields are refreshed every step)
        self.gcn_desc = [(layers.GinLayerDesc() if self.kind == "gin" else layers.GcnLayerDesc()) for _ in self.gcn]
        self.conv_api = "gt_gin_layer" if self.kind == "gin" else "gt_gcn_layer"
        self.vn_desc = [layers.VnUpdateDesc() for _ in self.vn]
        self.enc_desc = [layers.EncoderLayerDesc() for _ in self.enc_layers]
        self._fill_static()
        # the virtual-node update of layer l only feeds layer l+1: it runs on a second stream beside layer
        # l's conv (forward) / beside layer l's BatchNorm + aggregate backward (backward)
        self.side = _side_stream(self.dev, "vn") if (self.has_vn and OVERLAP_VN) else None
        # weight-gradient GEMMs run on a third stream beside the dX chain (gt_overlap_dw_*)
        self.side_dw = _side_stream(self.dev, "dw") if OVERLAP_DW else None
        lib = _lib.lib()
        nev = len(self.vn) if self.side is not None else 0
        self.ev_x = [lib.gt_event_create() for _ in range(nev)]      # x_l ready (main -> side)
        self.ev_vn = [lib.gt_event_create() for _ in range(nev)]     # vn_{l+1} ready (side -> main)
        self.ev_dvn = [lib.gt_event_create() for _ in range(nev)]    # d vn_{l+1} complete (main -> side)
        self.ev_extra = [lib.gt_event_create() for _ in range(nev)]  # d x_l extra complete (side -> main)
        self.ev_pool = [lib.gt_event_create() for _ in range(self.L if self.side is not None else 0)]   # d x_l complete (main -> side)
        self.ev_vnemb = [lib.gt_event_create()] if self.side is not None else []                        # d vn_0 reduced (side -> main)
        # the node-id sort for the embedding backward runs beside the forward on the dW stream
        self.ev_sort = [lib.gt_event_create(), lib.gt_event_create()] if self.side_dw is not None else []
        self.ev_wt = [lib.gt_event_create(), lib.gt_event_create()] if self.side_dw is not None else []   # transposed weights ready
        self.embed_sorted = bool(self.embed) and max(int(t.shape[0]) for t in self.embed) <= 16384
        # big-M fp32 GEMM weights that can run as bf16x6 on the bf16 matrix pipe (w3.py / csrc/linear3x.h): images built per step
        self.w3_weights = []
        for conv, _bn in self.gcn:
            if self.kind == "gcn":
                self.w3_weights.append(conv.linear.weight)
            else:
                m = list(conv.mlp)
                self.w3_weights += [m[0].weight, m[3].weight]
        self.w3_weights.append(g2t.weight)
        if self.embed_kind == "linear" and self.ne_Kp == self.ne_K:
            self.w3_weights.append(self.ne_lin.weight)
        self.w3_enc_weights = []   # the encoder layers' GEMMs run in fp32 only in the fp32 mode (fp32 token rows)
        for mod in self.enc_layers:
            self.w3_enc_weights += [mod.self_attn.in_proj_weight, mod.self_attn.out_proj.weight, mod.linear1.weight, mod.linear2.weight]
        self._w3 = {}   # with_encoder -> (W3Images, versions)
        # x_0 = h_0 + vn_0[batch] with vn_0 = the ONE row of virtualnode_embedding for every graph (gnn_module.py:195):
        # the embedding-sum kernel takes it as one more table whose index column is a stride-0 zero
        self.vn0_in_embed = self.has_vn and self.embed_kind != "linear" and len(self.embed) < 16
        self.zero_i64 = torch.zeros(1, dtype=torch.int64, device=self.dev)

    def __del__(self):
        try:
            lib = _lib.lib()
            for ev in self.ev_x + self.ev_vn + self.ev_dvn + self.ev_extra + self.ev_sort + self.ev_wt + self.ev_pool + self.ev_vnemb:
                lib.gt_event_destroy(ev)
        except Exception:
            pass

    def _fill_static(self):
        D = self.D
        for l, ((conv, bn), desc, mode) in enumerate(zip(self.gcn, self.gcn_desc, self.gcn_edge)):
            desc.D = D
            if self.kind == "gcn":
                desc.lin_w, desc.lin_b, desc.root = conv.linear.weight.data_ptr(), conv.linear.bias.data_ptr(), conv.root_emb.weight.data_ptr()
            else:
                m = list(conv.mlp)
                desc.eps = conv.eps.data_ptr()
                desc.w1, desc.b1, desc.bn1_w, desc.bn1_b = m[0].weight.data_ptr(), m[0].bias.data_ptr(), m[1].weight.data_ptr(), m[1].bias.data_ptr()
                desc.w2, desc.b2 = m[3].weight.data_ptr(), m[3].bias.data_ptr()
                desc.bn1_rm, desc.bn1_rv, desc.bn1_nbt = m[1].running_mean.data_ptr(), m[1].running_var.data_ptr(), m[1].num_batches_tracked.data_ptr()
            if mode == "linear":
                desc.edge_mode = GT_EDGE_LINEAR
                desc.edge_cols = conv.edge_encoder.weight.shape[1]
                desc.edge_w, desc.edge_b = conv.edge_encoder.weight.data_ptr(), conv.edge_encoder.bias.data_ptr()
            elif mode == "tables":
                desc.edge_mode = GT_EDGE_TABLES
                desc.edge_cols, desc.table_rows = len(self.tables[l]), self.table_rows[l]
                for i, o in enumerate(self.tab_off[l]):
                    desc.tab_off[i] = o
            else:
                desc.edge_mode = GT_EDGE_NONE
            desc.bn_w, desc.bn_b = bn.weight.data_ptr(), bn.bias.data_ptr()
            desc.bn_rm, desc.bn_rv, desc.bn_nbt = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()
            desc.bn_momentum, desc.bn_eps = float(bn.momentum), float(bn.eps)
        for m, desc in zip(self.vn, self.vn_desc):
            desc.D = D
            for name, p in zip(("w1", "b1", "bn1_w", "bn1_b", "w2", "b2", "bn2_w", "bn2_b"),
                               (m[0].weight, m[0].bias, m[1].weight, m[1].bias, m[3].weight, m[3].bias, m[4].weight, m[4].bias)):
                setattr(desc, name, p.data_ptr())
            desc.bn1_rm, desc.bn1_rv, desc.bn1_nbt = m[1].running_mean.data_ptr(), m[1].running_var.data_ptr(), m[1].num_batches_tracked.data_ptr()
            desc.bn2_rm, desc.bn2_rv, desc.bn2_nbt = m[4].running_mean.data_ptr(), m[4].running_var.data_ptr(), m[4].num_batches_tracked.data_ptr()
            desc.bn_momentum, desc.bn_eps = float(m[1].momentum), float(m[1].eps)
        for mod, desc in zip(self.enc_layers, self.enc_desc):
            desc.d_model, desc.ffn = self.d, mod.linear1.weight.shape[0]
            for name, p in zip(("in_w", "in_b", "out_w", "out_b", "l1_w", "l1_b", "l2_w", "l2_b", "n1_w", "n1_b", "n2_w",
                                "n2_b"), layers.encoder_layer_params(mod)):
                setattr(desc, name, p.data_ptr())
            desc.ln_eps = float(mod.norm1.eps)
            desc.act = self.enc_act

    def w3_images(self, with_encoder, stream):
        """bf16x3 images of the GEMM weights, rebuilt (one launch) whenever a weight changed since they were built"""
        from . import w3
        ws = self.w3_weights + (self.w3_enc_weights if with_encoder else [])
        ent = self._w3.get(with_encoder)
        if ent is None or not ent[0].current():
            ent = self._w3[with_encoder] = [w3.W3Images(ws), None]
        vers = (w3.EPOCH,) + tuple(w._version for w in ws)
        if ent[1] != vers:
            ent[0].build(stream)
            ent[1] = vers
        return ent[0]

    def w1_images(self, stream):
        """fragment-order bf16 images of the encoder layers' weights (w3.W1Images / csrc/linear1.h), rebuilt when a weight changed"""
        from . import w3
        ws = self.w3_enc_weights
        ent = self._w3.get("w1")
        if ent is None or not ent[0].current():
            ent = self._w3["w1"] = [w3.W1Images(ws), None]
        vers = (w3.EPOCH,) + tuple(w._version for w in ws)
        if ent[1] != vers:
            ent[0].build(stream)
            ent[1] = vers
        return ent[0]

    def small(self, B):
        c = self._cache.get(B)
        if c is None:
            c = dict(zeros=torch.zeros(B, dtype=torch.int32, device=self.dev),
                     ident=torch.arange(B, dtype=torch.int32, device=self.dev),
                     ptr01=torch.tensor([0, B], dtype=torch.int32, device=self.dev))
            self._cache[B] = c
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
    """True when the fused path covers this model / call (cached per model and mode)."""
    if perturb is not None or not getattr(model, "fused", True):
        return False
    # the fused node differentiates EVERY parameter: any frozen parameter (epoch_callback's freeze_gnn, or a user's
    # requires_grad_(False) on any submodule) sends the model through the module path.  The cache key holds every
    # parameter's requires_grad flag, so freezing anything after the first forward invalidates the cached answer.
    st = state(model)
    plist = st.get("params")
    if plist is None:
        plist = st["params"] = list(model.parameters())
    key = (model.training, torch.is_grad_enabled(), tuple(p.requires_grad for p in plist))
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
    """True when `model` is the `.module` of a torch DistributedDataParallel instance.  DDP reduces gradients from hooks on the
    parameters' AccumulateGrad nodes (C++ side, invisible from here), which a node that assigns `.grad` directly never reaches:
    such a model runs the module-by-module path (autograd accumulates as usual), or -- the supported way -- uses
    graphtrans_amd.dist.GradSync, whose all-reduce the fused backward issues itself.  Found through the wrapper that holds the
    model (wrapper.__dict__['_modules']['module'] is model)."""
    import gc
    from torch.nn.parallel import DistributedDataParallel
    for mods in gc.get_referrers(model):
        if not isinstance(mods, dict) or mods.get("module") is not model:
            continue
        for wd in gc.get_referrers(mods):
            if isinstance(wd, dict) and wd.get("_modules") is mods:
                if any(isinstance(o, DistributedDataParallel) for o in gc.get_referrers(wd)):
                    return True
    return False


def _eligible_static(model):
    from . import ops
    from .modules.conv import GCNConv
    from .modules.norm import BatchNorm1d
    if has_grad_hooks(model) or wrapped_in_ddp(model):   # (the fused node bypasses autograd's per-parameter machinery)
        return False
    gnn, enc = model.gnn_node, model.transformer_encoder
    try:
        if not model._use_packed() or gnn.JK not in ("last", "cat"):
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


# ---------------------------------------------------------------------------------------------------
# the autograd node
# ---------------------------------------------------------------------------------------------------
def _call(name, *args):
    _lib.check(getattr(_lib.lib(), name)(*args), name)


class _FusedModel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trigger, model, batched_data, gs, lay):
        from . import ops, w3
        plan = _plan(model)
        # exact-fp32 GEMM mode: the big-M linears run as bf16x6 on the bf16 matrix pipe (fp32-accurate, csrc/linear3x.h) on images
        # of their weights -- rebuilt here when a weight changed (one launch), bound for this host thread while the pass runs
        imgs = None
        if w3.ENABLED and ops.get_matmul_dtype() != torch.bfloat16 and gs.N >= 1024:
            imgs = plan.w3_images(model.transformer_encoder.compute_dtype != torch.bfloat16, _stream())
            imgs.bind()
        # bf16 token rows: the encoder layers' GEMMs run with the weight stationary in registers on fragment-order images (linear1.h)
        imgs1 = None
        ev_w1 = None
        prep = _PREPS.get(plan.dev.index or 0)
        prep_active = prep is not None and prep.active and getattr(gs, "ready_event", None) is not None   # begun for THIS call's structure
        if prep is not None:
            prep.active = False   # (consumed here whether or not anything follows the structure onto the side stream)
        if w3.W1_ENABLED and model.transformer_encoder.compute_dtype == torch.bfloat16 and plan.w3_enc_weights:
            if prep_active:   # this step's structure is being built on the side stream: the images follow it there
                imgs1 = plan.w1_images(prep.stream)
                ev_w1 = prep.w1_done()
            else:
                imgs1 = plan.w1_images(_stream())
            imgs1.bind()
        ctx.w1 = imgs1
        ctx.ev_w1 = ev_w1
        hook = _bn_sync_hook(model, plan) if model.training else None
        if hook is not None:
            hook.install()
        try:
            return _FusedModel._forward_body(ctx, model, batched_data, gs, lay, plan, imgs, hook)
        finally:
            if imgs is not None:
                imgs.unbind()
            if imgs1 is not None:
                imgs1.unbind()
            if hook is not None:
                hook.uninstall()
                if hook.error is not None:
                    e, hook.error = hook.error, None
                    raise e

    @staticmethod
    def _forward_body(ctx, model, batched_data, gs, lay, plan, imgs, hook=None):
        from . import ops
        L, D, d, dev = plan.L, plan.D, plan.d, plan.dev
        N, E, B, rows = gs.N, gs.E, gs.B, lay.rows
        st = _stream()
        lib = _lib.lib()
        training = 1 if model.training else 0
        # compute type of the fp32-stored GEMMs (message passing, gnn2transformer, heads): exact-fp32 MFMA or bf16 MFMA
        # (ops.set_matmul_dtype); the encoder layers compute in bf16 whenever their token rows are stored in bf16
        # (layers.hip: dtype == bf16 ? bf16 : compute), i.e. (fp32, bf16 tokens) is the mixed mode of bench.py
        compute = GT_BF16 if ops.get_matmul_dtype() == torch.bfloat16 else GT_F32
        enc = model.transformer_encoder
        tdt = GT_BF16 if enc.compute_dtype == torch.bfloat16 else GT_F32
        tsz = 2 if tdt == GT_BF16 else 4
        sm = plan.small(B)
        nenc = len(plan.enc_layers)

        # dropout seeds: drawn in the module path's order (GNN first, then the encoder)
        from .modules.gnn_module import _gnn_seed, layer_seed, vn_seed
        gnn_base = _gnn_seed(model.gnn_node)
        gnn_p = float(model.gnn_node.drop_ratio) if model.training else 0.0
        # ---- refresh the batch-dependent descriptor fields
        ea = batched_data.edge_attr
        ea_f = None
        if "linear" in plan.gcn_edge:
            ea_f = ea if (ea.dtype == torch.float32 and ea.is_contiguous()) else ea.float().contiguous()
        elif "tables" in plan.gcn_edge:
            if ea.dtype != torch.int64:
                raise TypeError("embedding-table edge encoders need int64 edge_attr")
            ea_f = ea.contiguous()
        for l, desc in enumerate(plan.gcn_desc):
            desc.N, desc.E, desc.B = N, E, B
            desc.has_vn = 1 if plan.has_vn else 0
            desc.relu = 1 if l != L - 1 else 0
            desc.residual = 1 if model.gnn_node.residual else 0
            desc.training, desc.compute = training, compute
            desc.dropout_p, desc.seed = gnn_p, layer_seed(gnn_base, l)
            layers._fill_graph(desc, gs)
            if plan.gcn_edge[l] and E > 0:
                desc.edge_attr = ea_f.data_ptr()
            ov = plan.side is not None and l < L - 1
            desc.ev_x_ready = None   # (x_l exists before the layer starts: the event is recorded from here, see below)
            desc.ev_dx_wait = plan.ev_extra[l] if ov else None
            # the virtual-node add of layer l+1 (h_list[l+1] + vn[batch], gnn_module.py:199) rides in layer l's BatchNorm
            # apply pass: layer l+1 then finds x_{l+1} ready (x_has_vn) -- no N x D read-modify-write pass per layer
            desc.x_has_vn = 1 if plan.has_vn else 0
            desc.vn_next, desc.ev_vn_next = None, None   # (pointers are arena-relative: filled below)
        for l, desc in enumerate(plan.vn_desc):
            desc.dropout_p, desc.seed = gnn_p, vn_seed(gnn_base, l)
            desc.N, desc.B = N, B
            desc.residual = 1 if model.gnn_node.residual else 0
            desc.training, desc.compute = training, compute
            desc.graph_ptr, desc.node_graph, desc.identity_graph = gs.graph_ptr.data_ptr(), gs.node_graph.data_ptr(), sm["ident"].data_ptr()
            desc.ev_dx_done = plan.ev_extra[l] if (plan.side is not None and VN_DEFER_DW) else None   # recorded inside the backward composite, ahead of its dW GEMMs
        p_drop = float(enc.dropout_p) if model.training else 0.0
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) if (model.training and enc.dropout_p > 0) else 0
        for i, desc in enumerate(plan.enc_desc):
            desc.rows, desc.nhead = rows, enc.nhead
            desc.dtype, desc.compute, desc.training = tdt, compute, training
            desc.seq_desc, desc.num_seqs, desc.row_stride, desc.max_npos = lay.desc.data_ptr(), lay.B, lay.row_stride, lay.max_npos
            work = getattr(lay, "work", None)
            desc.work_items, desc.num_work = (work.data_ptr() if work is not None else None), getattr(lay, "num_work", 0)
            desc.dropout_p = p_drop
            desc.seed = (seed + 0x9E3779B97F4A7C15 * (i + 1)) & 0xFFFFFFFFFFFFFFFF

        # ---- arena layout
        b = _Bump()
        ND4 = N * D * 4
        o = dict(h=[b.take(ND4) for _ in range(L + 1)])
        if plan.has_vn:
            o["x"] = [b.take(ND4 if (l == 0 and not plan.vn0_in_embed) else 0) for l in range(L)]
            o["vn"] = [b.take(B * D * 4) for _ in range(L)]
            vn_saved_bytes = [lib.gt_vn_update_saved_bytes(C.byref(dsc)) for dsc in plan.vn_desc]
            o["vn_saved"] = [b.take(n) for n in vn_saved_bytes]
        gcn_saved_bytes = [getattr(lib, plan.conv_api + "_saved_bytes")(C.byref(dsc)) for dsc in plan.gcn_desc]
        o["gcn_saved"] = [b.take(n) for n in gcn_saved_bytes]
        Kc = 2 * D if plan.jk_cat else D
        # JK = "cat" (modules/gnn_module.py:104-105): with bound weight images the gnn2transformer GEMM reads [h_list[0] | h_list[-1]]
        # from the two matrices where they lie and its backward writes the two gradients where their consumers read them
        cat2 = bool(plan.jk_cat and imgs is not None and lib.gt_linear_cat2_ok(compute, plan.g2t.weight.data_ptr(), N, d, D, D))
        if plan.jk_cat and not cat2:
            o["cat"] = b.take(N * Kc * 4)
        o["hn"] = b.take(N * d * tsz)
        o["tok"] = b.take(rows * d * tsz)
        if plan.norm_in is not None:
            o["x0"] = b.take(rows * d * tsz)
            o["st0"] = b.take(2 * rows * 4)
        o["xe"] = [b.take(rows * d * tsz) for _ in range(nenc)]
        enc_saved_bytes = [lib.gt_encoder_layer_saved_bytes(C.byref(dsc)) for dsc in plan.enc_desc]
        o["enc_saved"] = [b.take(n) for n in enc_saved_bytes]
        if plan.norm_out is not None:
            o["xo"] = b.take(rows * d * tsz)
            o["sto"] = b.take(2 * rows * 4)
        tab_rows_total = sum(plan.table_rows)
        o["etab"] = b.take(tab_rows_total * D * 4)
        will_bwd = bool(ctx.needs_input_grad[0])
        esort = will_bwd and plan.embed_sorted and plan.embed_kind != "linear"
        if esort:   # sorted node ids per table row (kept for the backward) + the sort's own scratch
            emb_rows_c = (C.c_int64 * len(plan.embed))(*[t.shape[0] for t in plan.embed])
            eplan_bytes = lib.gt_embed_sort_plan_bytes(len(plan.embed), emb_rows_c, N)
            esort_ws_bytes = lib.gt_embed_sort_workspace_bytes(len(plan.embed), emb_rows_c, N)
            o["eplan"] = b.take(eplan_bytes)
            o["esort_ws"] = b.take(esort_ws_bytes)
        if plan.embed_kind == "linear" and plan.ne_Kp != plan.ne_K:   # K-padded copies of x and W (16-byte chunks)
            o["ne_x"] = b.take(N * plan.ne_Kp * 4)
            o["ne_w"] = b.take(D * plan.ne_Kp * 4)
        o["hg"] = b.take(B * d * 4)
        o["wcat"] = b.take(0 if plan.head_w_flat is not None else plan.Nh * d * 4)
        o["bcat"] = b.take(0 if plan.head_w_flat is not None else plan.Nh * 4)
        ws_bytes = max([getattr(lib, plan.conv_api + "_workspace_bytes")(C.byref(dsc)) for dsc in plan.gcn_desc]
                       + [lib.gt_vn_update_workspace_bytes(C.byref(dsc)) for dsc in plan.vn_desc] + [256])
        o["ws"] = b.take(ws_bytes)
        ws2_bytes = max([lib.gt_vn_update_workspace_bytes(C.byref(dsc)) for dsc in plan.vn_desc] + [256])
        o["ws2"] = b.take(ws2_bytes)   # the side stream's workspace
        # transposed copies of the message-passing weights for the backward's exact-fp32 dX GEMMs (they run the forward-form
        # kernel on W^T): written once, beside the forward, instead of one transpose launch in front of every dX GEMM
        # (with bound images the dX GEMMs run on the image of W^T; only the BatchNorm-statistics epilogue of models without a
        # virtual node still takes the exact-fp32 kernel and its W^T)
        want_wt = will_bwd and compute == GT_F32 and N >= 1024 and (imgs is None or (not plan.has_vn and plan.kind == "gcn"))
        if want_wt:
            o["wt"] = [b.take((2 if plan.kind == "gin" else 1) * 2 * D * D * 4) for _ in range(L)]
            o["g2t_wt"] = b.take(d * Kc * 4)
        # a device-built token layout only has an upper bound on the row count: zero-filled buffers keep the rows past the
        # true count finite (they contribute exactly 0 to every weight gradient)
        arena = (torch.empty if lay.exact else torch.zeros)(b.off, dtype=torch.uint8, device=dev)
        base = arena.data_ptr()
        side = plan.side.cuda_stream if plan.side is not None else None

        def P(key, i=None):
            return base + (o[key] if i is None else o[key][i])

        for dsc in plan.gcn_desc:
            if plan.kind == "gin":
                dsc.w1_t, dsc.w2_t = None, None
            else:
                dsc.lin_wt = None
        g2t_wt = None
        if want_wt:
            tst = st
            if plan.side_dw is not None:
                tst = plan.side_dw.cuda_stream
                _call("gt_event_record", plan.ev_wt[0], st)       # the weights are as the optimizer left them on the main stream
                _call("gt_stream_wait_event", tst, plan.ev_wt[0])
            for l, dsc in enumerate(plan.gcn_desc):
                if plan.kind == "gin":   # w1 [2D][D], w2 [D][2D]
                    dsc.w1_t, dsc.w2_t = P("wt", l), P("wt", l) + 2 * D * D * 4
                    _call("gt_transpose", dsc.w1, dsc.w1_t, 2 * D, D, tst)
                    _call("gt_transpose", dsc.w2, dsc.w2_t, D, 2 * D, tst)
                else:
                    dsc.lin_wt = P("wt", l)
                    _call("gt_transpose", dsc.lin_w, dsc.lin_wt, D, D, tst)
            g2t_wt = P("g2t_wt")
            _call("gt_transpose", plan.g2t.weight.data_ptr(), g2t_wt, d, Kc, tst)
            if plan.side_dw is not None:
                _call("gt_event_record", plan.ev_wt[1], tst)
        if tab_rows_total:   # every layer's bond tables stacked by one copy: layer l reads its [rows_l][D] slice
            tv = arena[o["etab"]:o["etab"] + tab_rows_total * D * 4].view(torch.float32).view(tab_rows_total, D)
            torch.cat([t.detach() for tl in plan.tables for t in tl], out=tv)
            roff = 0
            for l, dsc in enumerate(plan.gcn_desc):
                if plan.gcn_edge[l] == "tables":
                    dsc.edge_w = P("etab") + roff * D * 4
                    roff += plan.table_rows[l]
        # ---- input encoder: h0 = sum of embedding rows   (dataset/utils.py:28-30 / ogb AtomEncoder)
        x = batched_data.x
        T = len(plan.embed)
        ne_x = ne_w = None
        if plan.embed_kind == "linear":   # h0 = x W^T + b on the MFMA GEMM (K zero-padded to 16-byte chunks)
            x = x.contiguous()
            nl, K, Kp = plan.ne_lin, plan.ne_K, plan.ne_Kp
            ne_x, ne_w = x.data_ptr(), nl.weight.data_ptr()
            if Kp != K:
                ne_x, ne_w = P("ne_x"), P("ne_w")
                _call("gt_repitch", ne_x, Kp, x.data_ptr(), K, N, 4, st)
                _call("gt_repitch", ne_w, Kp, nl.weight.data_ptr(), K, D, 4, st)
            _call("gt_linear_fwd", GT_F32, GT_F32, compute, ne_x, ne_w, nl.bias.data_ptr(), P("h", 0), N, D, Kp, 0, 0.0, 0, st)
            e_idx = e_str = e_clamp = cols = None
        elif plan.embed_kind == "ast":
            depth = batched_data.node_depth.reshape(-1)
            cols = [(x.data_ptr(), x.stride(0)), (x.data_ptr() + 8 * x.stride(1), x.stride(0)), (depth.data_ptr(), depth.stride(0) if N > 1 else 1)]
        else:
            cols = [(x.data_ptr() + 8 * i * x.stride(1), x.stride(0)) for i in range(T)]
        if plan.embed_kind != "linear":
            I64, PT = C.c_int64 * T, C.c_void_p * T
            e_idx, e_str = PT(*[c[0] for c in cols]), I64(*[c[1] for c in cols])
            e_clamp = I64(*plan.embed_clamp)
            e_tabs = PT(*[t.data_ptr() for t in plan.embed])
            if plan.vn0_in_embed:   # + virtualnode_embedding.weight[0] for every node
                I64f, PTf = C.c_int64 * (T + 1), C.c_void_p * (T + 1)
                _call("gt_embed_sum_fwd", T + 1, PTf(*[c[0] for c in cols], plan.zero_i64.data_ptr()), I64f(*[c[1] for c in cols], 0),
                      I64f(*plan.embed_clamp, -1), PTf(*[t.data_ptr() for t in plan.embed], plan.vn_emb.data_ptr()), N, D, P("h", 0), st)
            else:
                _call("gt_embed_sum_fwd", T, e_idx, e_str, e_clamp, e_tabs, N, D, P("h", 0), st)
            if esort:
                sst = st
                if plan.side_dw is not None:   # beside the forward: only the index columns are read
                    sst = plan.side_dw.cuda_stream
                    _call("gt_event_record", plan.ev_sort[0], st)
                    _call("gt_stream_wait_event", sst, plan.ev_sort[0])
                _call("gt_embed_sort", T, e_idx, e_str, e_clamp, emb_rows_c, N, P("eplan"), eplan_bytes, P("esort_ws"),
                      esort_ws_bytes, sst)
                if plan.side_dw is not None:
                    _call("gt_event_record", plan.ev_sort[1], sst)

        # ---- the graph structure may still be in the making on the side stream (Prep): GCN layer 0 waits between its GEMM and its
        # aggregate (descriptor), the virtual-node stream before its first segment sum; every other configuration right here
        ev_graph, gs.ready_event = getattr(gs, "ready_event", None), None
        late_wait = ev_graph is not None and plan.kind == "gcn" and (not plan.has_vn or plan.vn0_in_embed)
        if plan.kind == "gcn":
            plan.gcn_desc[0].ev_graph_ready = ev_graph if late_wait else None
        if ev_graph is not None:
            if not late_wait:
                _call("gt_stream_wait_event", st, ev_graph)
            elif side is not None:
                _call("gt_stream_wait_event", side, ev_graph)
        # ---- message passing   (modules/gnn_module.py:181-224)
        # x_l (= h_list[l] after its in-place virtual-node add): layer 0's comes from the embedding kernel (or a broadcast
        # add for Linear node encoders), layer l+1's is written by layer l's BatchNorm apply pass (vn_next)
        def X(l):
            if not plan.has_vn:
                return P("h", l)
            return P("x", 0) if (l == 0 and not plan.vn0_in_embed) else P("h", l)

        if plan.has_vn:
            _call("gt_segment_bcast_add", GT_F32, None, plan.vn_emb.data_ptr(), sm["zeros"].data_ptr(), B, 1, D, P("vn", 0), st)
        for l in range(L):
            dsc = plan.gcn_desc[l]
            if plan.has_vn:
                last = l == L - 1
                dsc.vn_next = None if last else P("vn", l + 1)
                dsc.ev_vn_next = plan.ev_vn[l] if (not last and side is not None) else None
                if l == 0 and not plan.vn0_in_embed:   # Linear node encoder: x_0 = h_0 + vn_0[batch] as its own pass
                    _call("gt_segment_bcast_add", GT_F32, P("h", 0), P("vn", 0), gs.node_graph.data_ptr(), N, B, D, P("x", 0), st)
                if not last and side is not None:
                    # the update of vn_{l+1} runs beside layer l's GEMM / aggregate on the second stream (it needs x_l,
                    # which exists when layer l starts: ev_x); layer l's apply pass waits for it (ev_vn_next)
                    _call("gt_event_record", plan.ev_x[l], st)
                    _call("gt_stream_wait_event", side, plan.ev_x[l])
                    _call("gt_vn_update_fwd", C.byref(plan.vn_desc[l]), X(l), P("vn", l), P("vn", l + 1), P("vn_saved", l),
                          P("ws2"), ws2_bytes, side)
                    _call("gt_event_record", plan.ev_vn[l], side)
                elif not last:
                    _call("gt_vn_update_fwd", C.byref(plan.vn_desc[l]), X(l), P("vn", l), P("vn", l + 1), P("vn_saved", l),
                          P("ws"), ws_bytes, st)
                _call(plan.conv_api + "_fwd", C.byref(dsc), X(l), P("vn", l), None, P("h", l + 1), P("gcn_saved", l), P("ws"),
                      ws_bytes, st)
            else:
                _call(plan.conv_api + "_fwd", C.byref(dsc), P("h", l), None, None, P("h", l + 1), P("gcn_saved", l), P("ws"),
                      ws_bytes, st)
        first = X(0)   # h_list[0] after the in-place virtual-node add
        g2t = plan.g2t
        if getattr(ctx, "ev_w1", None) is not None:   # the encoder's weight images were built on the side stream (Prep)
            _call("gt_stream_wait_event", st, ctx.ev_w1)
            ctx.ev_w1 = None
        if cat2:
            node_rep = None
            _call("gt_linear_fwd_cat2", tdt, compute, first, D, D, P("h", L), D, D, g2t.weight.data_ptr(), g2t.bias.data_ptr(), P("hn"),
                  N, d, d, st)
        else:
            if plan.jk_cat:   # torch.cat([h_list[0], h_list[-1]], 1)   (gnn_module.py:104-105)
                _call("gt_copy2d", P("cat"), Kc * 4, first, D * 4, D * 4, N, st)
                _call("gt_copy2d", P("cat") + D * 4, Kc * 4, P("h", L), D * 4, D * 4, N, st)
                node_rep = P("cat")
            else:
                node_rep = P("h", L)
            # ---- gnn2transformer + token rows + encoder   (models/gnn_transformer.py:92-114)
            _call("gt_linear_fwd", GT_F32, tdt, compute, node_rep, g2t.weight.data_ptr(), g2t.bias.data_ptr(), P("hn"), N, d, Kc,
                  0, 0.0, 0, st)
        cls_t = None
        if plan.cls is not None:
            cls_t = plan.cls.detach().reshape(-1)
            if tdt == GT_BF16:
                cls_t = cls_t.to(torch.bfloat16)
        _call("gt_seq_gather", tdt, P("hn"), None if cls_t is None else cls_t.data_ptr(), gs.graph_ptr.data_ptr(),
              lay.desc.data_ptr(), lay.B, lay.row_stride, lay.max_npos, 1 if lay.with_cls else 0, d, P("tok"), None, st)
        cur = P("tok")
        if plan.norm_in is not None:
            ln = plan.norm_in
            _call("gt_layernorm_fwd", tdt, cur, None, ln.weight.data_ptr(), ln.bias.data_ptr(), float(ln.eps), 0.0, 0, rows, d,
                  P("x0"), P("st0"), P("st0") + rows * 4, st)
            cur = P("x0")
        enc_in = []
        for i, dsc in enumerate(plan.enc_desc):
            enc_in.append(cur)
            _call("gt_encoder_layer_fwd", C.byref(dsc), cur, P("xe", i), P("enc_saved", i), st)
            cur = P("xe", i)
        pre_out = cur
        if plan.norm_out is not None:
            ln = plan.norm_out
            _call("gt_layernorm_fwd", tdt, cur, None, ln.weight.data_ptr(), ln.bias.data_ptr(), float(ln.eps), 0.0, 0, rows, d,
                  P("xo"), P("sto"), P("sto") + rows * 4, st)
            cur = P("xo")
        _call("gt_rows_gather", tdt, cur, lay.last_rows.data_ptr(), B, d, P("hg"), st)

        # ---- prediction heads as one GEMM over the stacked weights   (gnn_transformer.py:120-126)
        if len(plan.heads) == 1:
            wcat, bcat = plan.heads[0].weight.data_ptr(), plan.heads[0].bias.data_ptr()
        elif plan.head_w_flat is not None:
            wcat, bcat = plan.head_w_flat.data_ptr(), plan.head_b_flat.data_ptr()
        else:
            ww = arena[o["wcat"]:o["wcat"] + plan.Nh * d * 4].view(torch.float32).view(plan.Nh, d)
            wb = arena[o["bcat"]:o["bcat"] + plan.Nh * 4].view(torch.float32)
            torch.cat([h.weight.detach() for h in plan.heads], out=ww)
            torch.cat([h.bias.detach() for h in plan.heads], out=wb)
            wcat, bcat = P("wcat"), P("bcat")
        logits = torch.empty((B, plan.ldy), dtype=torch.float32, device=dev)
        _call("gt_linear_fwd_ld", GT_F32, GT_F32, compute, P("hg"), wcat, bcat, logits.data_ptr(), B, plan.Nh, d, plan.ldy, 0,
              0.0, 0, st)

        # the plan's descriptors are rewritten by the next forward: the backward gets its own copies
        snap = lambda ds: [type(x_).from_buffer_copy(x_) for x_ in ds]
        ctx.state = dict(gcn_desc=snap(plan.gcn_desc), vn_desc=snap(plan.vn_desc), enc_desc=snap(plan.enc_desc), plan=plan, arena=arena, o=o, base=base, gs=gs, lay=lay, sm=sm, compute=compute, tdt=tdt, tsz=tsz,
                         ws_bytes=ws_bytes, ws2_bytes=ws2_bytes, g2t_wt=g2t_wt, xptr=[X(l) for l in range(L)], enc_in=enc_in, pre_out=pre_out, first=first, node_rep=node_rep, Kc=Kc,
                         embed=(T, e_idx, e_str, e_clamp, cols), esort=esort, ne=(ne_x, ne_w), wcat=wcat, keep=(x, ea_f, cls_t, batched_data),
                         dims=(N, E, B, rows), sync=state(model).get("sync"), w3=imgs, bn_hook=hook, cat2=cat2, h_last=P("h", L))
        ctx.set_materialize_grads(False)
        if esort and plan.side_dw is not None:   # long finished; joins the sort's stream before anything can free the arena
            _call("gt_stream_wait_event", st, plan.ev_sort[1])
        elif want_wt and plan.side_dw is not None:
            # the transposed weights were written into THIS arena on the overlap stream: join it here too (the sort's event
            # above is recorded behind the transposes on the same in-order stream), so that an autograd graph dropped
            # without a backward cannot hand the arena back to the allocator under pending side-stream writes
            _call("gt_stream_wait_event", st, plan.ev_wt[1])
        out = logits[:, :plan.Nh] if plan.ldy != plan.Nh else logits
        return out

    @staticmethod
    def backward(ctx, dlogits):
        s = ctx.state
        if s is None:
            raise RuntimeError("graphtrans_amd fused model: backward through the graph a second time "
                               "(the saved activations are freed after the first backward)")
        plan, o, base, gs, lay, sm = s["plan"], s["o"], s["base"], s["gs"], s["lay"], s["sm"]
        compute, tdt, tsz = s["compute"], s["tdt"], s["tsz"]
        N, E, B, rows = s["dims"]
        L, D, d, dev, Kc = plan.L, plan.D, plan.d, plan.dev, s["Kc"]
        st = _stream()
        lib = _lib.lib()
        from . import ops
        if dlogits is None:
            return None, None, None, None, None
        dl = ops._padded_rows(dlogits.reshape(B, plan.Nh).to(torch.float32), plan.ldy) if plan.ldy != plan.Nh \
            else dlogits.reshape(B, plan.Nh).to(torch.float32).contiguous()

        def P(key, i=None):
            return base + (o[key] if i is None else o[key][i])

        # gradients: straight into the persistent flat buffer when nothing has to be accumulated
        direct = all(p.grad is None for p in plan.plist)
        model_sync = s["sync"]
        flat = plan.flat if direct else torch.empty_like(plan.flat)
        G = flat.data_ptr()

        # ---- backward arena
        nenc = len(s["enc_desc"])
        b = _Bump()
        q = dict(d_hg=b.take(B * d * 4), dtok=[b.take(rows * d * tsz) for _ in range(2)], d_hn=b.take(N * d * tsz),
                 d_cls=b.take(B * d * tsz), d_rep=b.take(N * Kc * 4), dA=b.take(N * D * 4), dB=b.take(N * D * 4),
                 dC=b.take(N * D * 4), dJ=b.take(N * D * 4 if plan.jk_cat else 0), dvn=[b.take(B * D * 4) for _ in range(4)])
        enc_ws = max([lib.gt_encoder_layer_workspace_bytes(C.byref(dsc)) for dsc in s["enc_desc"]] + [256])
        ln_ws = lib.gt_layernorm_bwd_workspace_bytes(rows, d)
        lin_ws = max(lib.gt_linear_bwd_workspace_bytes(compute, B, plan.Nh, d), lib.gt_linear_bwd_workspace_bytes(compute, N, d, Kc))
        emb_rows = (C.c_int64 * len(plan.embed))(*[t.shape[0] for t in plan.embed])
        if plan.embed_kind == "linear":
            emb_ws = lib.gt_linear_bwd_workspace_bytes(compute, N, D, plan.ne_Kp)
            q["ne_dw"] = b.take(D * plan.ne_Kp * 4 if plan.ne_Kp != plan.ne_K else 0)
        elif s["esort"]:
            emb_ws = lib.gt_embed_sum_bwd_sorted_workspace_bytes(len(plan.embed), N, D)
        else:
            emb_ws = lib.gt_embed_sum_bwd_workspace_bytes(len(plan.embed), emb_rows, D)
        ws_bytes = max(s["ws_bytes"], enc_ws, ln_ws, lin_ws, emb_ws)
        # BatchNorm-backward statistics summed in the dX epilogue of the layer above (GCN, exact-fp32 GEMMs, no GNN dropout):
        # per-64-row-tile partial rows, one buffer per BatchNorm that is fed by a conv's dX (all but the last layer's)
        d0 = s["gcn_desc"][0]
        # (with a virtual node the main stream waits for the virtual-node chain between two dX GEMMs anyway: measured 0.4 %
        # slower there -- the longer dX epilogue delays that chain -- so only models without one take it)
        fuse_bn = (s.get("bn_hook") is None and not plan.has_vn and plan.kind == "gcn" and bool(d0.training) and d0.dropout_p == 0.0 and
                   bool(lib.gt_linear_bwd_bnstats_ok(compute, GT_F32, GT_F32, N)))
        s["fuse_bn"] = fuse_bn
        s["bn_rows"] = int(lib.gt_linear_bwd_bnstats_rows(N)) if fuse_bn else 0
        q["bnpart"] = [b.take(s["bn_rows"] * 2 * D * 4 if fuse_bn else 0) for _ in range(max(L - 1, 0))]
        s["heads_ws_bytes"] = int(lib.gt_linear_bwd_workspace_bytes(compute, B, plan.Nh, d))
        q["heads_ws"] = b.take(s["heads_ws_bytes"])   # the heads' dW partials: their own buffer (see the heads stage below)
        q["ws"] = [b.take(ws_bytes), b.take(ws_bytes)]   # alternated between consecutive stages (see W() below)
        q["ws2"] = b.take(s["ws2_bytes"])
        seg_ws_bytes = lib.gt_segment_sum_workspace_bytes(N, D) if plan.has_vn else 0
        q["ws3"] = b.take(seg_ws_bytes)   # the per-graph pooling of d x_l runs on the second stream with its own scratch
        s["seg_ws_bytes"] = seg_ws_bytes
        barena = (torch.empty if lay.exact else torch.zeros)(b.off, dtype=torch.uint8, device=dev)
        bb = barena.data_ptr()
        side = plan.side.cuda_stream if plan.side is not None else None

        def Q(key, i=None):
            return bb + (q[key] if i is None else q[key][i])

        # the overlap stream pays for itself only when the kernels are long enough to hide its extra stream operations
        # (~4 per GEMM): on NCI1-sized batches (1 k nodes) the step is host-bound and it cost 40 %
        ov = plan.side_dw is not None and N * D >= DW_OVERLAP_MIN_ELEMS
        s["ov"] = ov
        dw_sync = (lambda: _call("gt_overlap_dw_sync")) if ov else (lambda: None)
        if ov:
            _call("gt_overlap_dw_begin", st, plan.side_dw.cuda_stream)
        if s.get("w3") is not None:
            s["w3"].bind()   # (autograd's worker thread: the table is per host thread)
        w1 = getattr(ctx, "w1", None)
        if w1 is not None:
            w1.bind()
        if s.get("bn_hook") is not None:
            s["bn_hook"].install()
        try:
            return _FusedModel._backward_body(ctx, s, plan, o, q, P, Q, G, flat, dl, direct, model_sync, ws_bytes, dw_sync, barena,
                                              emb_rows, st)
        finally:
            if ov:
                _lib.lib().gt_overlap_dw_end()
            if s.get("w3") is not None:
                s["w3"].unbind()
            if w1 is not None:
                w1.unbind()
            if s.get("bn_hook") is not None:
                s["bn_hook"].uninstall()
                if s["bn_hook"].error is not None:
                    e, s["bn_hook"].error = s["bn_hook"].error, None
                    raise e

    @staticmethod
    def _backward_body(ctx, s, plan, o, q, P, Q, G, flat, dl, direct, model_sync, ws_bytes, dw_sync, barena, emb_rows, st):
        gs, lay, sm = s["gs"], s["lay"], s["sm"]
        compute, tdt, tsz = s["compute"], s["tdt"], s["tsz"]
        N, E, B, rows = s["dims"]
        L, D, d, dev, Kc = plan.L, plan.D, plan.d, plan.dev, s["Kc"]
        nenc = len(s["enc_desc"])
        side = plan.side.cuda_stream if plan.side is not None else None
        ov = s["ov"]
        slot = [0]

        def W(join=False):
            """the next stage's workspace.  The two slots alternate, so the weight-gradient GEMMs a stage forks onto the
            third stream (their partials and the dy they read live in the stage's slot) can run beside the NEXT stage: the
            main stream then waits only for the GEMMs that used this slot two stages ago (gt_overlap_dw_release).
            join=True waits for all of them.  (The encoder composite forks its last dW (in_proj) ahead of its dX GEMM and the
            heads fork theirs first: leftovers of the hunt for the irreproducible LayerNorm backward, DESIGN.md section 8 --
            the cause was an inline-asm conversion, not the schedule; the order costs nothing and stays.)"""
            slot[0] ^= 1
            p = Q("ws", slot[0])
            if ov:
                if join:
                    _call("gt_overlap_dw_sync")
                else:
                    _call("gt_overlap_dw_release", p, ws_bytes)
            return p

        # ---- heads
        # the weight gradient first (256 x 25 010 x 128: 85 us): forked onto the overlap stream it runs beside the heads' own dX
        # GEMM and the stages after it, in a buffer of its own -- the dX GEMM's split-N partials take the stage's workspace
        _call("gt_linear_bwd_dw_forked", GT_F32, GT_F32, compute, P("hg"), s["wcat"], dl.data_ptr(), None, G + plan.headw_off * 4,
              G + plan.headb_off * 4, B, plan.Nh, d, d, plan.ldy, 0.0, Q("heads_ws"), s["heads_ws_bytes"], st)
        _call("gt_linear_bwd_ld", GT_F32, GT_F32, compute, P("hg"), s["wcat"], dl.data_ptr(), None, None, None, Q("d_hg"),
              None, None, B, plan.Nh, d, plan.ldy, 0.0, W(), ws_bytes, st)
        # ---- pooled rows -> token rows
        dcur, dnext = Q("dtok", 0), Q("dtok", 1)
        _call("gt_rows_scatter", tdt, Q("d_hg"), lay.last_rows.data_ptr(), B, rows, d, dcur, st)
        if plan.norm_out is not None:
            ln = plan.norm_out
            _call("gt_layernorm_bwd", tdt, s["pre_out"], None, dcur, ln.weight.data_ptr(), P("sto"), P("sto") + rows * 4, 0.0, 0,
                  rows, d, dnext, None, G + plan.norm_out_off[0] * 4, G + plan.norm_out_off[1] * 4, W(), ws_bytes, st)
            dcur, dnext = dnext, dcur
        for i in range(nenc - 1, -1, -1):
            _call("gt_encoder_layer_bwd", C.byref(s["enc_desc"][i]), s["enc_in"][i], dcur, P("enc_saved", i), dnext,
                  G + plan.enc_off[i] * 4, W(), ws_bytes, st)
            dcur, dnext = dnext, dcur
        if plan.norm_in is not None:
            ln = plan.norm_in
            _call("gt_layernorm_bwd", tdt, P("tok"), None, dcur, ln.weight.data_ptr(), P("st0"), P("st0") + rows * 4, 0.0, 0,
                  rows, d, dnext, None, G + plan.norm_in_off[0] * 4, G + plan.norm_in_off[1] * 4, W(), ws_bytes, st)
            dcur, dnext = dnext, dcur
        # ---- token rows -> node rows (+ the CLS gradient)
        _call("gt_seq_scatter", tdt, dcur, None, gs.graph_ptr.data_ptr(), gs.node_graph.data_ptr(), lay.desc.data_ptr(), lay.B,
              lay.row_stride, 1 if lay.with_cls else 0, N, d, Q("d_hn"), Q("d_cls") if plan.cls is not None else None, st)
        if plan.cls is not None:
            dc = barena[q["d_cls"]:q["d_cls"] + B * d * tsz].view(torch.bfloat16 if tdt == GT_BF16 else torch.float32).view(B, d)
            torch.sum(dc, dim=0, dtype=torch.float32, out=flat[plan.cls_off:plan.cls_off + d])
        g2t = plan.g2t
        if s["g2t_wt"] is not None and plan.ev_wt:
            _call("gt_stream_wait_event", st, plan.ev_wt[1])   # the transposed weights were written on the overlap stream beside the forward
        if s["cat2"]:   # d h_list[0] -> dJ, d h_list[-1] -> dA straight from the GEMM (no d_rep, no copies)
            _call("gt_linear_bwd_cat2", tdt, compute, s["first"], D, D, s["h_last"], D, D, g2t.weight.data_ptr(), Q("d_hn"), Q("dJ"), D,
                  Q("dA"), D, G + plan.g2t_off[0] * 4, G + plan.g2t_off[1] * 4, N, d, d, W(), ws_bytes, st)
        else:
            _call("gt_linear_bwd_wt", GT_F32, tdt, compute, s["node_rep"], g2t.weight.data_ptr(), s["g2t_wt"], Q("d_hn"), None, None, None,
                  Q("d_rep"), G + plan.g2t_off[0] * 4, G + plan.g2t_off[1] * 4, N, d, Kc, 0.0, W(), ws_bytes, st)
        # every gradient from gnn2transformer onwards is final: put that half of the flat buffer on the wire
        sync = model_sync if (direct and model_sync is not None and model_sync.active) else None
        if sync is not None:
            dw_sync()
            sync.reduce_flat(flat, plan.g2t_off[0], plan.total)
        # ---- message passing, last layer first.  dy = d h_list[l+1]; "extra" = gradient reaching x_l (=
        # h_list[l] after the virtual-node add) from its consumers other than conv_l: the JK slab (l = 0)
        # and the virtual-node update's pooling (l < L-1).
        if s["cat2"]:
            dy = Q("dA")
        elif plan.jk_cat:
            _call("gt_copy2d", Q("dA"), D * 4, Q("d_rep") + D * 4, Kc * 4, D * 4, N, st)   # d h_list[-1]
            _call("gt_copy2d", Q("dJ"), D * 4, Q("d_rep"), Kc * 4, D * 4, N, st)           # d h_list[0]
            dy = Q("dA")
        else:
            dy = Q("d_rep")
        if s["fuse_bn"]:
            for l in range(1, L):
                up, dn = s["gcn_desc"][l], s["gcn_desc"][l - 1]
                up.prev_saved = P("gcn_saved", l - 1)
                up.prev_bn_w, up.prev_bn_b = dn.bn_w, dn.bn_b
                up.prev_relu = dn.relu
                up.prev_bn_part = Q("bnpart", l - 1)
                dn.bn_part_in, dn.bn_nparts_in = Q("bnpart", l - 1), s["bn_rows"]
        d_vn_next = None   # total gradient of vn_{l+1}
        for l in range(L - 1, -1, -1):
            extra = Q("dJ") if (l == 0 and plan.jk_cat) else None
            upd = plan.has_vn and l < L - 1
            if upd:   # vn_{l+1} = update(x_l, vn_l): d x_l = pooled gradient (+ the JK slab at l = 0)
                if side is not None:   # beside layer l's BatchNorm / aggregate backward; joined before its dX GEMM
                    _call("gt_event_record", plan.ev_dvn[l], st)
                    _call("gt_stream_wait_event", side, plan.ev_dvn[l])
                    _call("gt_vn_update_bwd", C.byref(s["vn_desc"][l]), d_vn_next, P("vn_saved", l), extra, Q("dC"), Q("dvn", 2),
                          G + plan.vn_off[l] * 4, Q("ws2"), s["ws2_bytes"], side)   # records ev_extra[l] itself (ev_dx_done) ...
                    if not VN_DEFER_DW:
                        _call("gt_event_record", plan.ev_extra[l], side)
                else:
                    _call("gt_vn_update_bwd", C.byref(s["vn_desc"][l]), d_vn_next, P("vn_saved", l), extra, Q("dC"), Q("dvn", 2),
                          G + plan.vn_off[l] * 4, W(), ws_bytes, st)
                extra = Q("dC")
            out = Q("dB") if dy == Q("dA") else Q("dA")
            xin = s["xptr"][l]
            if l == 0 and ov:
                _call("gt_overlap_dw_urgent", 1)   # layer 0's weight gradients are the last: nothing left to overlap them with
            pool_on_side = plan.has_vn and side is not None
            _call(plan.conv_api + "_bwd", C.byref(s["gcn_desc"][l]), xin, dy, extra, P("gcn_saved", l), out,
                  Q("dvn", 3) if (plan.has_vn and not pool_on_side) else None, G + plan.gcn_off[l] * 4, W(), ws_bytes, st)
            if plan.has_vn:   # d vn_l = (layer l's broadcast add: per-graph sum of d x_l) + (update l's pooled + residual inputs)
                # off the main chain: only the NEXT virtual-node update backward (second stream) reads it
                vst = side if pool_on_side else st
                if pool_on_side:
                    _call("gt_event_record", plan.ev_pool[l], st)
                    _call("gt_stream_wait_event", side, plan.ev_pool[l])
                    _call("gt_segment_sum_ws", GT_F32, out, None, gs.graph_ptr.data_ptr(), N, B, D, Q("dvn", 3), Q("ws3"),
                          s["seg_ws_bytes"], side)
                tgt = Q("dvn", l % 2)
                if upd:
                    _call("gt_segment_bcast_add", GT_F32, Q("dvn", 3), Q("dvn", 2), sm["ident"].data_ptr(), B, B, D, tgt, vst)
                else:
                    _call("gt_copy2d", tgt, D * 4, Q("dvn", 3), D * 4, D * 4, B, vst)
                d_vn_next = tgt
            dy = out
        d_h0 = dy
        if plan.has_vn:
            vst = side if side is not None else st
            _call("gt_segment_sum", GT_F32, d_vn_next, None, sm["ptr01"].data_ptr(), B, 1, D, G + plan.vn_emb_off * 4, vst)
            if side is not None:
                _call("gt_event_record", plan.ev_vnemb[0], side)
                if sync is not None:   # the gradient range below goes on the wire now
                    _call("gt_stream_wait_event", st, plan.ev_vnemb[0])
        # the message-passing gradients (everything between the embedding tables and gnn2transformer) are final:
        # on the wire while the embedding backward runs; the tables themselves follow right after it
        gnn_lo = plan.vn_emb_off if plan.has_vn else plan.gcn_off[0]
        if sync is not None:
            dw_sync()
            sync.reduce_flat(flat, gnn_lo, plan.g2t_off[0])
        # ---- input encoder tables
        if plan.embed_kind == "linear":   # dW = d_h0^T x, db = colsum(d_h0); the features need no gradient
            ne_x, ne_w = s["ne"]
            K, Kp = plan.ne_K, plan.ne_Kp
            dw = G + plan.ne_off[0] * 4 if Kp == K else Q("ne_dw")
            _call("gt_linear_bwd", GT_F32, GT_F32, s["compute"], ne_x, ne_w, d_h0, None, None, None, None, dw,
                  G + plan.ne_off[1] * 4, N, D, Kp, 0.0, W(), ws_bytes, st)
            dw_sync()
            if Kp != K:
                _call("gt_repitch", G + plan.ne_off[0] * 4, K, dw, Kp, D, 4, st)
        else:
            T, e_idx, e_str, e_clamp, _cols = s["embed"]
            d_tabs = (C.c_void_p * T)(*[G + off * 4 for off in plan.embed_off])
            if s["esort"]:
                _call("gt_embed_sum_bwd_sorted", T, emb_rows, d_h0, N, D, P("eplan"), d_tabs, W(), ws_bytes, st)
            else:
                _call("gt_embed_sum_bwd", T, e_idx, e_str, e_clamp, emb_rows, d_h0, N, D, d_tabs, W(), ws_bytes, st)

        if sync is not None:
            sync.reduce_flat(flat, 0, gnn_lo)
        elif plan.has_vn and side is not None:
            _call("gt_stream_wait_event", st, plan.ev_vnemb[0])   # joined only here: the embedding backward ran beside the tail of the virtual-node chain
        # ---- hand the gradients to the parameters
        if not direct:
            # accumulation reads `flat` on the main stream right here: every fork that still writes it (layer 0's urgent dW
            # GEMMs, aggregate partial reduces, LayerNorm column finishes on the overlap stream) has to be joined first (the virtual-node stream was joined just above) -- the `finally` join of gt_overlap_dw_end comes too late
            dw_sync()
        if direct:
            for p, v in zip(plan.plist, plan.views):
                p.grad = v
        else:
            off_views = [flat[o_:o_ + p.numel()].view(p.shape) for p, o_ in plan.params]
            for p, v in zip(plan.plist, off_views):
                if p.grad is None:
                    p.grad = v
                else:
                    p.grad = p.grad + v
        ctx.state = None
        return None, None, None, None, None


def forward(model, batched_data, gs, lay):
    """logits (B, Nh) [row-padded storage] of the fused path; `model` must be `eligible`."""
    plan = _plan(model)
    return _FusedModel.apply(plan.plist[0], model, batched_data, gs, lay)
