"""oracle/make_golden.py — generate tests/golden/*.npz by running the REFERENCE ITSELF.

*** TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference). ***

The reference's own modules (`modules/conv.py`, `modules/gnn_module.py`, `modules/utils.py`,
`modules/transformer_encoder.py`, `modules/masked_transformer_encoder.py`,
`models/gnn_transformer.py`, `dataset/utils.py`) are imported UNMODIFIED from /root/reference;
their missing third-party imports are satisfied by our stubs in oracle/stubs/ (see its README).
No reference source is copied: the fixtures are data (inputs, state_dict, outputs, gradients).

Usage:  python oracle/make_golden.py            # rewrites tests/golden/*.npz
Each fixture: in.* inputs, sd.* state_dict, out.* outputs, gin.* input grads, gsd.* param grads
for L = sum_i sum(out_i * w_i) with fixed random w (in.w<i>), fp32, fixed seeds, dropout 0.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "stubs"))


def _bind_reference_package(name):
    """Register /root/reference/<name> under the dotted name `<name>` BEFORE anything imports it.
    The repository root carries alias packages of the same names (`models/`, `modules/`: the
    drop-in boundary); the reference's `modules/` has no __init__.py, i.e. it is a namespace package,
    and a regular package anywhere on sys.path beats a namespace package whatever the path order.
    Binding by explicit file path makes the fixtures come from the REFERENCE's classes whatever
    sys.path holds; `_assert_reference_bound` checks it after the imports."""
    import types

    path = os.path.join(REF, name)
    init = os.path.join(path, "__init__.py")
    if os.path.exists(init):
        spec = importlib.util.spec_from_file_location(name, init, submodule_search_locations=[path])
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    else:
        mod = types.ModuleType(name)
        mod.__path__ = [path]
        sys.modules[name] = mod
    return mod


for _stale in [k for k in sys.modules if k.split(".")[0] in ("models", "modules", "dataset", "trainers")]:
    del sys.modules[_stale]
sys.path.insert(0, REF)     # the reference's flat imports (`import utils`, `from trainers import ...`)
_bind_reference_package("modules")   # namespace package in the reference
_bind_reference_package("models")    # (must precede modules.transformer_encoder: circular import, main.py:21)
if REPO not in sys.path:
    sys.path.append(REPO)   # graphtrans_amd.synth / oracle.* only; `models` / `modules` are already bound above

import models  # noqa: E402,F401
from models.gnn_transformer import GNNTransformer  # noqa: E402
from modules.conv import GCNConv, GINConv  # noqa: E402
from modules.gnn_module import GNNNodeEmbedding  # noqa: E402
from modules.masked_transformer_encoder import Block, CausalSelfAttention  # noqa: E402
from modules.transformer_encoder import TransformerNodeEncoder  # noqa: E402
from modules.utils import pad_batch, unpad_batch  # noqa: E402
from ogb.graphproppred.mol_encoder import AtomEncoder, BondEncoder  # noqa: E402



def _assert_reference_bound():
    import inspect

    for cls in (GNNTransformer, GCNConv, GINConv, GNNNodeEmbedding, Block, CausalSelfAttention, TransformerNodeEncoder):
        f = os.path.realpath(inspect.getsourcefile(cls))
        assert f.startswith(REF + os.sep), f"{cls.__name__} was imported from {f}, not from the reference"
    for fn in (pad_batch, unpad_batch):
        f = os.path.realpath(inspect.getsourcefile(fn))
        assert f.startswith(REF + os.sep), f"{fn.__name__} was imported from {f}, not from the reference"


_assert_reference_bound()

_spec = importlib.util.spec_from_file_location("ref_dataset_utils", os.path.join(REF, "dataset/utils.py"))
_du = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_du)
ASTNodeEncoder = _du.ASTNodeEncoder

from graphtrans_amd import synth  # noqa: E402
from oracle.graph_struct import graph_struct, pad_index  # noqa: E402
from oracle.reference_math import default_args  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")

# torch 1.7.1 (the reference's pin, requirement.yml:37) always materialises a new tensor in
# F.dropout; torch 2.10 returns its input when p == 0 or in eval mode, which makes the reference's
# in-place residual `h += h_list[layer]` (modules/gnn_module.py:92-93) fail autograd on the ReLU
# output.  Restore the 1.7.1 aliasing behaviour (values unchanged) for fixture generation only.
import torch.nn.functional as _F  # noqa: E402

_orig_dropout = _F.dropout


def _dropout_fresh(input, p=0.5, training=True, inplace=False):
    out = _orig_dropout(input, p, training, inplace)
    return out.clone() if out is input else out


_F.dropout = _dropout_fresh


def zero_edge_encoder_cls(_):  # dataset/tud.py:67-71 behaviour
    def zero(_):
        return 0

    return zero


def edge_cls(kind):
    return {"linear": lambda d: nn.Linear(2, d), "bond": lambda d: BondEncoder(emb_dim=d),
            "none": zero_edge_encoder_cls}[kind]


def randomize(module, seed):
    """Give every parameter / buffer a non-trivial value (BN affine, root_emb, eps, VN emb are
    otherwise 1/0 and would hide errors)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 if p.dim() > 1 else 0.3))
            if name.endswith("norm1.weight") or name.endswith("norm2.weight") or "batch_norms" in name and name.endswith("weight") \
                    or name.endswith("norm.weight") or name.endswith("norm_input.weight") or name.endswith("ln1.weight") \
                    or name.endswith("ln2.weight") or (".mlp.1.weight" in name) or ("mlp_virtualnode_list" in name and (name.endswith(".1.weight") or name.endswith(".4.weight"))):
                p.add_(1.0)
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.2)
            elif name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)


def dump(name, meta, inputs, module, outs, grad_inputs, extra=None):
    d = {"meta": np.array(json.dumps(meta))}
    d.update(extra or {})
    for k, v in inputs.items():
        if v is not None:
            d["in." + k] = v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    if module is not None:
        for k, v in module.state_dict().items():
            if k.endswith("pos_encoder.pe"):  # deterministic sin/cos table, 5000 rows: not stored
                continue
            d["sd." + k] = v.detach().numpy()
        for k, p in module.named_parameters():
            if p.grad is not None:
                d["gsd." + k] = p.grad.detach().numpy()
    for i, o in enumerate(outs):
        d[f"out.{i}"] = o.detach().numpy()
    for k, v in grad_inputs.items():
        d["gin." + k] = v.detach().numpy()
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def run_and_dump(name, meta, inputs, module, fwd, float_inputs=()):
    """fwd() -> tensor or list of tensors.  Loss = sum_i (out_i * w_i).sum()."""
    g = torch.Generator().manual_seed(1234)
    for k in float_inputs:
        inputs[k].requires_grad_(True)
    outs = fwd()
    outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
    loss = 0
    for i, o in enumerate(outs):
        w = torch.randn(o.shape, generator=g)
        inputs[f"w{i}"] = w
        loss = loss + (o * w).sum()
    if loss.requires_grad:
        loss.backward()
    gin = {k: inputs[k].grad for k in float_inputs if inputs[k].grad is not None}
    dump(name, meta, inputs, module, outs, gin)


def batch_inputs(b):
    d = dict(x=b.x, edge_index=b.edge_index, batch=b.batch)
    if b.edge_attr is not None:
        d["edge_attr"] = b.edge_attr
    if hasattr(b, "node_depth"):
        d["node_depth"] = b.node_depth.clone()  # ASTNodeEncoder clamps its input in place (dataset/utils.py:29)
    return d


# ---------------------------------------------------------------------------------------------
def g1_g2_convs():
    D = 16
    for conv_name, cls in (("gcn", GCNConv), ("gin", GINConv)):
        for ek in ("linear", "bond", "none"):
            for mode in ("train", "eval"):
                if mode == "eval" and (conv_name == "gcn" or ek != "linear"):
                    continue
                torch.manual_seed(0)
                feat = "mol" if ek == "bond" else "dense"
                b = synth.tiny_mixed(seed=3, sizes=(14, 1, 23, 12), feat=feat, num_features=D)
                x = torch.randn(b.num_nodes, D)
                conv = cls(D, edge_cls(ek))
                randomize(conv, 7)
                conv.train(mode == "train")
                ea = b.edge_attr
                inputs = dict(x=x, edge_index=b.edge_index, edge_attr=ea, batch=b.batch)
                meta = dict(kind="conv", conv=conv_name, edge=ek, training=(mode == "train"), D=D)
                run_and_dump(f"G{1 if conv_name == 'gcn' else 2}_{conv_name}_{ek}_{mode}", meta, inputs, conv,
                             lambda: conv(x, b.edge_index, ea), float_inputs=("x",))


def make_node_encoder(feat, D):
    if feat == "code2":
        return ASTNodeEncoder(D, num_nodetypes=11, num_nodeattributes=13, max_depth=20)
    if feat == "mol":
        return AtomEncoder(D)
    if feat == "tud":
        return nn.Linear(6, D)
    raise ValueError


def g3_g4_gnn():
    D = 16
    grid = []
    for vn in (False, True):
        for jk in ("last", "sum", "cat"):
            for res in (False, True):
                grid.append((vn, jk, res))
    for i, (vn, jk, res) in enumerate(grid):
        gnn_type = "gcn" if i % 2 == 0 else "gin"
        feat, ek = [("code2", "linear"), ("mol", "bond"), ("tud", "none")][i % 3]
        for mode in ("train",) if i % 4 else ("train", "eval"):
            torch.manual_seed(0)
            b = synth.tiny_mixed(seed=10 + i, sizes=(9, 1, 17, 6), feat=feat)
            args = default_args(gnn_virtual_node=vn, gnn_num_layer=3, gnn_emb_dim=D, gnn_JK=jk, gnn_residual=res,
                                gnn_type=gnn_type, gnn_dropout=0.0)
            m = GNNNodeEmbedding(vn, 3, D, make_node_encoder(feat, D), edge_cls(ek), JK=jk, drop_ratio=0.0,
                                 residual=res, gnn_type=gnn_type)
            randomize(m, 20 + i)
            m.train(mode == "train")
            perturb = torch.randn(b.num_nodes, D) * 0.1 if i % 5 == 0 else None
            inputs = batch_inputs(b)
            if perturb is not None:
                inputs["perturb"] = perturb
            meta = dict(kind="gnn_node", args=vars(args), training=(mode == "train"), feat=feat, edge=ek)
            tag = f"G{4 if vn else 3}_{gnn_type}_{'vn_' if vn else ''}{jk}_{'res' if res else 'nores'}_{feat}_{mode}"
            run_and_dump(tag, meta, inputs, m, lambda: m(b, perturb),
                         float_inputs=(("x",) if feat == "tud" else ()) + (("perturb",) if perturb is not None else ()))


def g5_pad():
    for name, sizes, max_len in (("ragged", (5, 1, 9, 3), 1000), ("trunc", (5, 1, 9, 3, 12), 7), ("equal", (4, 4, 4), 4)):
        torch.manual_seed(0)
        batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
        h = torch.randn(batch.numel(), 8, requires_grad=True)
        padded, mask, num_nodes, masks, S = pad_batch(h, batch, max_len, get_mask=True)
        # pad_batch writes through .data (modules/utils.py:17) so autograd sees it via the indexed copy
        prev = torch.randn(batch.numel(), 8)
        pin = torch.randn(S, len(sizes), 8)
        unp = unpad_batch(pin, prev, num_nodes, masks, S)
        g = torch.Generator().manual_seed(5)
        w = torch.randn(padded.shape, generator=g)
        (padded * w).sum().backward()
        d = {"meta": np.array(json.dumps(dict(kind="pad", max_input_len=max_len))),
             "in.h": h.detach().numpy(), "in.batch": batch.numpy(), "in.w0": w.numpy(), "in.prev": prev.numpy(),
             "in.padded_in": pin.numpy(), "out.0": padded.detach().numpy(), "out.1": mask.numpy(),
             "out.2": unp.numpy(), "gin.h": h.grad.numpy()}
        os.makedirs(OUT, exist_ok=True)
        np.savez_compressed(os.path.join(OUT, f"G5_pad_{name}.npz"), **d)
        print(f"G5_pad_{name}")


def ragged_padded(sizes, S, d, seed):
    g = torch.Generator().manual_seed(seed)
    B = len(sizes)
    x = torch.zeros(S, B, d)
    mask = torch.zeros(B, S, dtype=torch.bool)
    for i, n in enumerate(sizes):
        x[S - n:, i] = torch.randn(n, d, generator=g)
        mask[i, : S - n] = True
    return x, mask


def g6_encoder():
    d = 16
    i = 0
    for cls_on in (True, False):
        for norm_input in (True, False):
            for act in ("relu", "gelu"):
                for mode in ("train", "eval") if (cls_on and norm_input) else ("train",):
                    torch.manual_seed(i)
                    args = default_args(d_model=d, nhead=2, dim_feedforward=24, transformer_dropout=0.0,
                                        transformer_activation=act, num_encoder_layers=2, max_input_len=1000,
                                        transformer_norm_input=norm_input, graph_pooling="cls" if cls_on else "mean")
                    m = TransformerNodeEncoder(args)
                    randomize(m, 40 + i)
                    m.train(mode == "train")
                    sizes = (7, 1, 11, 4)
                    x, mask = ragged_padded(sizes, 11, d, 50 + i)
                    inputs = dict(padded=x, mask=mask)
                    meta = dict(kind="transformer_node_encoder", args=vars(args), training=(mode == "train"))
                    run_and_dump(f"G6_enc_{'cls' if cls_on else 'nocls'}_{'ni' if norm_input else 'noni'}_{act}_{mode}",
                                 meta, inputs, m, lambda: m(x, mask)[0], float_inputs=("padded",))
                    i += 1


def g7_masked():
    d, nh, T, B = 16, 2, 9, 3
    i = 0
    for with_adj in (False, True):
        for with_valid in (False, True):
            torch.manual_seed(i)
            att = CausalSelfAttention(d, nh, 0.0, 0.0)
            randomize(att, 60 + i)
            x = torch.randn(B, T, d)
            adj = (torch.rand(B, T, T) < 0.4).float() if with_adj else None
            valid = torch.ones(B, T)
            valid[0, :3] = 0
            valid[2, :5] = 0
            valid = valid if with_valid else None
            inputs = dict(x=x)
            if adj is not None:
                inputs["attn_mask"] = adj
            if valid is not None:
                inputs["valid_input_mask"] = valid
            meta = dict(kind="causal_self_attention", n_head=nh)
            run_and_dump(f"G7_csa_{'adj' if with_adj else 'noadj'}_{'valid' if with_valid else 'novalid'}", meta, inputs,
                         att, lambda: att(x, adj, valid), float_inputs=("x",))
            i += 1
    for prenorm in (True, False):
        torch.manual_seed(i)
        blk = Block(d, 24, nh, 0.0, 0.0, prenorm=prenorm)
        randomize(blk, 70 + i)
        x = torch.randn(B, T, d)
        adj = (torch.rand(B, T, T) < 0.5).float()
        valid = torch.ones(B, T)
        valid[1, :4] = 0
        inputs = dict(x=x, attn_mask=adj, valid_input_mask=valid)
        meta = dict(kind="masked_block", n_head=nh, prenorm=prenorm)
        run_and_dump(f"G7_block_{'pre' if prenorm else 'post'}", meta, inputs, blk, lambda: blk(x, adj, valid),
                     float_inputs=("x",))
        i += 1


def g8_model():
    cases = [
        # name, feat, edge, args overrides, num_tasks, sizes
        ("c1_nci1", "tud", "none", dict(gnn_type="gcn", gnn_virtual_node=False, gnn_JK="last", graph_pooling="cls",
                                        gnn_num_layer=3, num_encoder_layers=3), 2, (9, 1, 17, 6)),
        ("c2_molpcba", "mol", "bond", dict(gnn_type="gin", gnn_virtual_node=True, gnn_JK="cat", graph_pooling="cls",
                                           transformer_norm_input=True, gnn_num_layer=3, num_encoder_layers=2), 6, (9, 2, 17, 6)),
        ("c3_code2", "code2", "linear", dict(gnn_type="gcn", gnn_virtual_node=True, gnn_JK="cat", graph_pooling="cls",
                                             transformer_norm_input=True, gnn_num_layer=3, num_encoder_layers=2, max_seq_len=5), 9,
         (9, 1, 17, 6)),
        ("c3_code2_trunc", "code2", "linear", dict(gnn_type="gcn", gnn_virtual_node=True, gnn_JK="cat", graph_pooling="cls",
                                                   transformer_norm_input=True, gnn_num_layer=2, num_encoder_layers=2,
                                                   max_seq_len=5, max_input_len=8), 9, (9, 1, 17, 6)),
        ("mean_pool", "code2", "linear", dict(gnn_type="gin", gnn_virtual_node=False, gnn_JK="sum", graph_pooling="mean",
                                              gnn_num_layer=2, num_encoder_layers=2), 4, (9, 3, 17, 6)),
        ("last_pos", "code2", "linear", dict(gnn_type="gcn", gnn_virtual_node=False, gnn_JK="last", graph_pooling="last",
                                             gnn_num_layer=2, num_encoder_layers=1, pos_encoder=True, gnn_residual=True), 4,
         (9, 1, 17, 6)),
        ("masked", "code2", "linear", dict(gnn_type="gcn", gnn_virtual_node=False, gnn_JK="last", graph_pooling="cls",
                                           gnn_num_layer=2, num_encoder_layers=1, num_encoder_layers_masked=2), 4, (9, 4, 17, 6)),
    ]
    D, d = 16, 16
    for i, (name, feat, ek, over, num_tasks, sizes) in enumerate(cases):
        for mode in ("train", "eval") if name.startswith("c3_code2") else ("train",):
            torch.manual_seed(100 + i)
            args = default_args(gnn_emb_dim=D, d_model=d, nhead=2, dim_feedforward=24, transformer_dropout=0.0,
                                gnn_dropout=0.0, **over)
            b = synth.tiny_mixed(seed=200 + i, sizes=sizes, feat=feat)
            if args.num_encoder_layers_masked > 0:
                adj_list = []
                ptr = np.concatenate([[0], np.cumsum(sizes)])
                ei = b.edge_index.numpy()
                for g in range(len(sizes)):
                    n = sizes[g]
                    a = np.zeros((n, n), dtype=bool)
                    sel = (ei[0] >= ptr[g]) & (ei[0] < ptr[g + 1])
                    a[ei[0][sel] - ptr[g], ei[1][sel] - ptr[g]] = True
                    adj_list.append(a)
                b.adj_list = adj_list
            m = GNNTransformer(num_tasks, make_node_encoder(feat, D), edge_cls(ek), args)
            randomize(m, 300 + i)
            m.train(mode == "train")
            inputs = batch_inputs(b)
            if hasattr(b, "adj_list"):
                for g, a in enumerate(b.adj_list):
                    inputs[f"adj{g}"] = a
            meta = dict(kind="gnn_transformer", args=vars(args), training=(mode == "train"), feat=feat, edge=ek,
                        num_tasks=num_tasks, sizes=list(sizes))
            run_and_dump(f"G8_{name}_{mode}", meta, inputs, m, lambda: m(b), float_inputs=("x",) if feat == "tud" else ())


def g9_pna():
    """In-tree PNA aggregators / scalers (modules/pna/aggregators.py, scalers.py) on hand-made
    segments incl. empty ones; the only runnable pieces of the reference's PNA statement."""
    spec = importlib.util.spec_from_file_location("ref_pna_aggr", os.path.join(REF, "modules/pna/aggregators.py"))
    aggr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(aggr)
    spec = importlib.util.spec_from_file_location("ref_pna_scal", os.path.join(REF, "modules/pna/scalers.py"))
    scal = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scal)
    g = torch.Generator().manual_seed(9)
    n, e = 9, 40
    index = torch.randint(0, n, (e,), generator=g)
    index[index == 4] = 5  # segment 4 is empty; segment 8 possibly too
    index[index == 8] = 0
    src = torch.randn(e, 4, 6, generator=g)
    d = {"meta": np.array(json.dumps(dict(kind="pna_aggr", n=n))), "in.src": src.numpy(), "in.index": index.numpy()}
    for name in ("mean", "max", "min", "std", "var", "sum"):
        d["out." + name] = aggr.AGGREGATORS[name](src, index, n).numpy()
    deg = torch.bincount(index, minlength=n).float().view(-1, 1, 1)
    avg = {"lin": 2.5, "log": 1.1, "exp": 20.0}
    d["in.deg"] = deg.numpy()
    x = torch.randn(n, 4, 6, generator=g)
    d["in.x"] = x.numpy()
    for name in ("identity", "amplification", "attenuation", "linear", "inverse_linear"):
        d["out.scale_" + name] = scal.SCALERS[name](x.clone(), deg, avg).numpy()
    np.savez_compressed(os.path.join(OUT, "G9_pna_aggr_scalers.npz"), **d)
    print("G9_pna_aggr_scalers")


def _load_ref_pna():
    """The reference's OWN statement of PNAConv (modules/pna_layer.py:20-171) is dead code in the tree: its ctor uses
    the bare names ModuleList / Sequential / ReLU / Linear (`:102-118`) that the file never imports, so constructing
    it raises NameError.  The four names are bound in that module's namespace here (to torch.nn's classes, what PyG
    1.6.3's own `pna_conv.py` imports); the class body -- ctor, forward (`:131-146`), message (`:148-160`),
    aggregate (`:162-168`) -- runs UNMODIFIED.  modules/pna/pna_module.py then imports
    `torch_geometric.nn.PNAConv` (third party, absent): that name is bound to this in-tree class, which has PyG's
    constructor signature, so PNANodeEmbedding / PNATransformer run unmodified too."""
    import importlib

    import torch_geometric.nn as tgnn

    pl = importlib.import_module("modules.pna_layer")
    for name in ("ModuleList", "Sequential", "ReLU", "Linear"):
        if not hasattr(pl, name):
            setattr(pl, name, getattr(nn, name))
    tgnn.PNAConv = pl.PNAConv
    pm = importlib.import_module("modules.pna.pna_module")
    pm.PNAConv = pl.PNAConv
    pt = importlib.import_module("models.pna_transformer")
    return pl.PNAConv, pm.PNANodeEmbedding, pt.PNATransformer


PNA_DEG_HIST = [0, 41, 30, 12, 6, 3, 1, 1]   # in-degree histogram (dataset/code.py:122-130 builds an 800-bin one)


def run_and_dump64(name, meta, inputs, module, fwd, float_inputs=()):
    """Like run_and_dump, plus the SAME module evaluated in float64 (parameters and float inputs cast up from
    their fp32 values): out64.* / gin64.* / gsd64.* give the conditioning-free answer of the reference's own code.
    `fwd(module, inputs)` -> tensor or list of tensors."""
    import copy

    g = torch.Generator().manual_seed(1234)
    for k in float_inputs:
        inputs[k].requires_grad_(True)
    outs = fwd(module, inputs)
    outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
    loss = 0
    for i, o in enumerate(outs):
        w = torch.randn(o.shape, generator=g)
        inputs[f"w{i}"] = w
        loss = loss + (o * w).sum()
    loss.backward()
    gin = {k: inputs[k].grad for k in float_inputs if inputs[k].grad is not None}
    m64 = copy.deepcopy(module).double()
    m64.zero_grad()
    in64 = {k: (v.detach().double().requires_grad_(k in float_inputs) if isinstance(v, torch.Tensor) and v.is_floating_point() else v)
            for k, v in inputs.items()}
    outs64 = fwd(m64, in64)
    outs64 = list(outs64) if isinstance(outs64, (list, tuple)) else [outs64]
    sum((o * in64[f"w{i}"]).sum() for i, o in enumerate(outs64)).backward()
    extra = {}
    for i, o in enumerate(outs64):
        extra[f"out64.{i}"] = o.detach().numpy()
    for k, p in m64.named_parameters():
        if p.grad is not None:
            extra["gsd64." + k] = p.grad.detach().numpy()
    for k in float_inputs:
        if in64[k].grad is not None:
            extra["gin64." + k] = in64[k].grad.numpy()
    dump(name, meta, inputs, module, outs, gin, extra)


class _B:   # attribute bag standing in for a PyG Batch
    def __init__(self, **kw):
        self.__dict__.update(kw)


def g12_pna():
    """G12: pins the PNA conv wiring (a11).  PNAConv = the reference's in-tree class, PNANodeEmbedding /
    PNATransformer = the reference's modules, all unmodified (see _load_ref_pna).  Every case is stored twice: as the
    fp32 run, and as the float64 run of the same code and parameter values -- the fp32 `std` aggregator
    sqrt(relu(E[m^2] - E[m]^2) + 1e-5) (modules/pna/aggregators.py:27-34) cancels catastrophically for in-degree <= 1
    segments, so the fp32 reference's own gradients carry up to ~1e-3 relative rounding noise there."""
    PNAConv, PNANodeEmbedding, PNATransformer = _load_ref_pna()
    aggregators, scalers = ["mean", "max", "min", "std"], ["identity", "amplification", "attenuation"]
    deg = torch.tensor(PNA_DEG_HIST)
    # ---- the conv alone (towers=4, divide_input=True, no edge features: modules/pna/pna_module.py:43-51,73)
    for i, (D, sizes, aggs, scs) in enumerate(((16, (14, 1, 23, 12), aggregators, scalers),
                                               (32, (9, 2, 17), ["mean", "max", "min", "std"], ["identity", "linear", "inverse_linear"]),
                                               (16, (30,), ["max", "std"], ["attenuation"]))):
        torch.manual_seed(i)
        b = synth.tiny_mixed(seed=30 + i, sizes=sizes, feat="dense", num_features=D)
        conv = PNAConv(D, D, aggregators=aggs, scalers=scs, deg=deg, towers=4, divide_input=True)
        randomize(conv, 80 + i)
        inputs = dict(x=torch.randn(b.num_nodes, D), edge_index=b.edge_index, batch=b.batch)
        meta = dict(kind="pna_conv", D=D, towers=4, aggregators=aggs, scalers=scs, deg=PNA_DEG_HIST)
        run_and_dump64(f"G12_pnaconv_{i}", meta, inputs, conv, lambda m, a: m(a["x"], a["edge_index"]), float_inputs=("x",))

    def as_batch(a):
        return _B(x=a["x"], edge_index=a["edge_index"], edge_attr=a.get("edge_attr"), batch=a["batch"],
                  node_depth=a["node_depth"].clone())

    # ---- PNANodeEmbedding (modules/pna/pna_module.py:57-78)
    D = 16
    for i, (res, mode, with_perturb) in enumerate(((True, "train", False), (True, "eval", False), (False, "train", True))):
        torch.manual_seed(10 + i)
        args = default_args(gnn_num_layer=2, gnn_emb_dim=D, gnn_residual=res, gnn_dropout=0.0, graph_pooling="cls",
                            aggregators=aggregators, scalers=scalers, deg=deg)
        b = synth.tiny_mixed(seed=40 + i, sizes=(9, 1, 17, 6), feat="code2")
        m = PNANodeEmbedding(make_node_encoder("code2", D), args)
        randomize(m, 90 + i)
        m.train(mode == "train")
        inputs = batch_inputs(b)
        if with_perturb:
            inputs["perturb"] = torch.randn(b.num_nodes, D) * 0.1
        meta = dict(kind="pna_node", args=dict(vars(args), deg=PNA_DEG_HIST), training=(mode == "train"), feat="code2")
        run_and_dump64(f"G12_pnanode_{'res' if res else 'nores'}_{mode}", meta, inputs, m,
                       lambda mm, a: mm(as_batch(a), a.get("perturb")), float_inputs=("perturb",) if with_perturb else ())
    # ---- PNATransformer end to end (models/pna_transformer.py:78-100)
    for i, (pool, msl, mode) in enumerate((("cls", 5, "train"), ("cls", 5, "eval"), ("mean", None, "train"), ("last", None, "train"))):
        torch.manual_seed(20 + i)
        args = default_args(gnn_num_layer=2, gnn_emb_dim=D, gnn_residual=True, gnn_dropout=0.0, graph_pooling=pool,
                            d_model=16, nhead=2, dim_feedforward=24, transformer_dropout=0.0, num_encoder_layers=2,
                            transformer_norm_input=(pool == "cls"), max_seq_len=msl, aggregators=aggregators,
                            scalers=scalers, deg=deg, max_input_len=(12 if pool == "last" else 1000))
        b = synth.tiny_mixed(seed=50 + i, sizes=(9, 1, 17, 6), feat="code2")
        m = PNATransformer(7, make_node_encoder("code2", D), None, args)
        randomize(m, 95 + i)
        m.train(mode == "train")
        meta = dict(kind="pna_transformer", args=dict(vars(args), deg=PNA_DEG_HIST), training=(mode == "train"),
                    feat="code2", num_tasks=7)
        run_and_dump64(f"G12_pnatrans_{pool}_{mode}", meta, batch_inputs(b), m, lambda mm, a: mm(as_batch(a)))


def g10_struct():
    for name, b in (("tiny", synth.tiny_mixed(seed=3, sizes=(14, 1, 23, 12), feat="dense", num_features=4)),
                    ("code2", synth.code2_like(B=6, seed=1)), ("mol", synth.molpcba_like(B=12, seed=2))):
        s = graph_struct(b.edge_index.numpy(), b.batch.numpy())
        S, kept, first = pad_index(s["ptr"], 20)
        d = {"meta": np.array(json.dumps(dict(kind="graph_struct", max_input_len=20))),
             "in.edge_index": b.edge_index.numpy(), "in.batch": b.batch.numpy(),
             "out.S": np.array(S), "out.kept": kept, "out.first": first}
        for k, v in s.items():
            d["out." + k] = v
        np.savez_compressed(os.path.join(OUT, f"G10_struct_{name}.npz"), **d)
        print(f"G10_struct_{name}")


def pack_store(graphs):
    """Flatten raw graphs into the store layout data.GraphStore uses (node_ptr/edge_ptr + concatenated arrays)."""
    d = {"node_ptr": np.concatenate([[0], np.cumsum([g["x"].shape[0] for g in graphs])]).astype(np.int64),
         "edge_ptr": np.concatenate([[0], np.cumsum([g["edge_index"].shape[1] for g in graphs])]).astype(np.int64),
         "x": np.concatenate([g["x"] for g in graphs], 0),
         "edge_index": np.concatenate([g["edge_index"] for g in graphs], 1)}
    for k in ("node_depth", "node_is_attributed", "edge_attr", "y", "y_arr"):
        if k in graphs[0]:
            d[k] = np.concatenate([g[k] for g in graphs], 0)
    return d


def g11_collate():
    """augment_edge (dataset/utils.py:89-141) run by the reference itself, per graph; the batch is then
    assembled by the PyG collation rule restated in oracle/collate.py (PyG is not importable here)."""
    from oracle import collate as oc

    class Data:  # augment_edge only touches attributes
        pass

    rng = np.random.default_rng(11)
    graphs = synth.code2_raw(B=5, seed=4, mean_nodes=20.0, max_nodes=60)
    one = dict(x=np.array([[3, 7]], np.int64), edge_index=np.zeros((2, 0), np.int64), node_depth=np.zeros((1, 1), np.int64),
               node_is_attributed=np.ones((1, 1), np.int64), y_arr=rng.integers(0, 50, (1, 5)))
    none_attr = {k: v.copy() for k, v in graphs[1].items()}
    none_attr["node_is_attributed"][:] = 0
    single_attr = {k: v.copy() for k, v in graphs[2].items()}
    single_attr["node_is_attributed"][:] = 0
    single_attr["node_is_attributed"][4] = 1
    graphs = graphs + [one, none_attr, single_attr]
    ids = np.array([3, 0, 6, 7, 5, 2, 1, 4], np.int64)
    aug = []
    for g in graphs:
        d = Data()
        d.edge_index = torch.from_numpy(g["edge_index"])
        d.node_is_attributed = torch.from_numpy(g["node_is_attributed"])
        d = _du.augment_edge(d)
        ei, ea = d.edge_index.numpy(), d.edge_attr.numpy()
        ei2, ea2 = oc.augment_edge(g["edge_index"], g["node_is_attributed"])
        assert ei.dtype == np.int64 and ea.dtype == np.float32
        assert np.array_equal(ei, ei2) and np.array_equal(ea, ea2), "oracle/collate.py disagrees with the reference"
        aug.append(dict(g, edge_index=ei, edge_attr=ea))
    out = oc.collate([aug[i] for i in ids])
    d = {"meta": np.array(json.dumps(dict(kind="collate", augment=True))), "in.ids": ids}
    d.update({"store." + k: v for k, v in pack_store(graphs).items()})
    d.update({"out." + k: v for k, v in out.items()})
    np.savez_compressed(os.path.join(OUT, "G11_collate_code2.npz"), **d)
    print("G11_collate_code2")

    graphs = synth.molpcba_raw(B=9, seed=5)
    ids = np.array([8, 2, 2, 0, 5], np.int64)
    out = oc.collate([graphs[i] for i in ids])
    d = {"meta": np.array(json.dumps(dict(kind="collate", augment=False))), "in.ids": ids}
    d.update({"store." + k: v for k, v in pack_store(graphs).items()})
    d.update({"out." + k: v for k, v in out.items()})
    np.savez_compressed(os.path.join(OUT, "G11_collate_mol.npz"), **d)
    print("G11_collate_mol")


if __name__ == "__main__":
    torch.set_num_threads(1)
    argv = sys.argv[1:]
    if "--out" in argv:     # write somewhere else (tests/test_golden_recipe.py regenerates into a tmp dir)
        i = argv.index("--out")
        OUT = os.path.abspath(argv[i + 1])
        del argv[i:i + 2]
        os.makedirs(OUT, exist_ok=True)
    if argv:   # regenerate selected groups only, e.g. `make_golden.py g11_collate`
        for name in argv:
            globals()[name]()
        sys.exit(0)
    g1_g2_convs()
    g3_g4_gnn()
    g5_pad()
    g6_encoder()
    g7_masked()
    g8_model()
    g9_pna()
    g10_struct()
    g11_collate()
    g12_pna()
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"total {total / 1024:.0f} KB in {len(os.listdir(OUT))} files")
