"""GPU parity of the PNA path (gt_pna_aggregate_fwd/bwd, PNAConv, PNANodeEmbedding, PNATransformer)
against (i) the G12 golden fixtures = the reference's own in-tree PNAConv (modules/pna_layer.py:131-167, run
unmodified) inside its unmodified PNANodeEmbedding / PNATransformer, and (ii) the CPU oracle (pinned to the same
fixtures by tests/test_oracle_golden.py) on seeded batches up to the C4 shape (D=272, L=4, 128 graphs).

The oracle side runs in float64: the reference formulation std = sqrt(relu(E[m^2] - E[m]^2) + 1e-5)
with m = U + V loses ~1 % of a small variance (1e-4) to fp32 cancellation, while the kernel's
shift-free E[V^2] - E[V]^2 matches the exact value; comparing both fp32 results would test that
noise, not the math.  Tolerance 1e-4 scale-relative against the float64 oracle."""
import numpy as np
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(seed, n_graphs=6, with_dupes=True):
    from graphtrans_amd import synth

    b = synth.tiny_mixed(seed=seed, sizes=tuple(int(s) for s in np.random.default_rng(seed).integers(1, 30, n_graphs)),
                         feat="code2")
    if with_dupes:  # duplicate some edges: ties for max/min between identical sources
        ei = b.edge_index
        b.edge_index = torch.cat([ei, ei[:, ::3]], dim=1).contiguous()
    return b


@pytest.mark.parametrize("D,towers", [(272, 4), (64, 4), (16, 1), (520, 2)])
def test_pna_aggregate_vs_oracle(D, towers):
    from graphtrans_amd import ops
    from graphtrans_amd.graph import GraphStructure
    from oracle import reference_math as rm

    torch.manual_seed(0)
    b = _graph(3)
    N = b.num_nodes
    F = D // towers
    U, V = torch.randn(N, D), torch.randn(N, D)
    w = torch.randn(N, towers, 4 * F)
    # oracle: m_k = U[dst] + V[src] per edge, then the in-tree aggregators
    Ur, Vr = U.double().requires_grad_(True), V.double().requires_grad_(True)
    row, col = b.edge_index[0], b.edge_index[1]
    m = (Ur[col] + Vr[row]).view(-1, towers, F)
    ref = rm.pna_aggregators(m, col, N, ["mean", "max", "min", "std"])
    (ref * w.double()).sum().backward()
    gs = GraphStructure.build(b.edge_index.to(DEV), b.batch.to(DEV), num_graphs=b.num_graphs)
    Ud, Vd = U.to(DEV).requires_grad_(True), V.to(DEV).requires_grad_(True)
    out = ops.pna_aggregate(Ud, Vd, gs, towers)
    (out * w.to(DEV)).sum().backward()
    assert_close(out.cpu(), ref.detach(), what="agg")
    assert_close(Ud.grad.cpu(), Ur.grad, what="dU")
    assert_close(Vd.grad.cpu(), Vr.grad, what="dV")


def _args(**kw):
    from oracle.reference_math import default_args

    return default_args(gnn_emb_dim=32, gnn_num_layer=2, gnn_residual=True, gnn_dropout=0.0, d_model=32, nhead=2,
                        dim_feedforward=48, transformer_dropout=0.0, num_encoder_layers=2, transformer_norm_input=True,
                        aggregators=["mean", "max", "min", "std"], scalers=["identity", "amplification", "attenuation"],
                        deg=torch.tensor([0, 5, 9, 7, 3, 1, 0, 2]), **kw)


def _randomize(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.2))
            if n.endswith("module.weight") or "norm" in n and n.endswith("weight"):
                p.add_(1.0)


@pytest.mark.parametrize("pooling,max_seq_len,training", [("cls", 3, True), ("mean", None, True), ("cls", None, False)])
def test_pna_transformer_vs_oracle(pooling, max_seq_len, training):
    from graphtrans_amd.encoders import ASTNodeEncoder
    from graphtrans_amd.models.pna_transformer import PNATransformer
    from oracle import reference_math as rm

    torch.manual_seed(1)
    args = _args(graph_pooling=pooling, max_seq_len=max_seq_len)
    b = _graph(5, with_dupes=False)
    model = PNATransformer(7, ASTNodeEncoder(32, 11, 13, 20), None, args)
    _randomize(model, 11)
    model.train(training)
    sd = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    torch.set_default_dtype(torch.float64)  # oracle helpers allocate zeros/ones in the default dtype
    try:
        ref = rm.pna_transformer(sd, args, b, None, training)
        ref = ref if isinstance(ref, list) else [ref]
        g = torch.Generator().manual_seed(2)
        ws = [torch.randn(r.shape, generator=g, dtype=torch.float64) for r in ref]
        sum((r * w).sum() for r, w in zip(ref, ws)).backward()
    finally:
        torch.set_default_dtype(torch.float32)
    ws = [w.float() for w in ws]
    model = model.to(DEV)
    out = model(b.to(DEV))
    out = out if isinstance(out, list) else [out]
    sum((o * w.to(DEV)).sum() for o, w in zip(out, ws)).backward()
    for i, (o, r) in enumerate(zip(out, ref)):
        assert_close(o.detach().cpu(), r.detach(), what=f"out{i}")
    for k, p in model.named_parameters():
        if sd[k].grad is not None:
            assert_close(p.grad.cpu(), sd[k].grad, what=f"grad {k}")


# ---- G12: the reference's own PNA code (fp32 run + float64 run of the same parameters).  The HIP path must be as close
# ---- to the float64 answer as 1e-4 or 3x the reference's own fp32 rounding noise, whichever is larger (the fp32 `std`
# ---- aggregator is ill-conditioned on in-degree <= 1 segments; see tests/test_oracle_golden.py).
def _close64(got, g, kind, key, what):
    ref = {"out": g.outs64, "gsd": g.gsd64, "gin": g.gin64}[kind][key]
    tol = max(1e-4, 3 * g.ref_noise(kind, key))
    assert_close(got.double().cpu(), ref, atol=tol, rtol=tol, what=what)


def _golden_args(g):
    a = g.args()
    a.deg = torch.tensor(a.deg)
    return a


from conftest import Golden, golden_names  # noqa: E402


@pytest.mark.parametrize("name", golden_names("G12_pnaconv"))
def test_pna_conv_golden(name):
    from graphtrans_amd.modules.pna.pna_module import PNAConv
    from helpers import load_sd

    g = Golden(name)
    m = g.meta
    conv = PNAConv(m["D"], m["D"], aggregators=m["aggregators"], scalers=m["scalers"], deg=torch.tensor(m["deg"]),
                   towers=m["towers"], divide_input=True)
    load_sd(conv, g.sd)
    conv = conv.to(DEV)
    x = g.inputs["x"].to(DEV).requires_grad_(True)
    out = conv(x, g.inputs["edge_index"].to(DEV))
    (out * g.inputs["w0"].to(DEV)).sum().backward()
    _close64(out.detach(), g, "out", "0", "out")
    _close64(x.grad, g, "gin", "x", "dx")
    for k, p in conv.named_parameters():
        _close64(p.grad, g, "gsd", k, f"grad {k}")


@pytest.mark.parametrize("name", golden_names("G12_pnanode"))
def test_pna_node_embedding_golden(name):
    from graphtrans_amd.modules.pna.pna_module import PNANodeEmbedding
    from helpers import load_sd, node_encoder

    g = Golden(name)
    args = _golden_args(g)
    model = PNANodeEmbedding(node_encoder("code2", args.gnn_emb_dim), args)
    load_sd(model, g.sd)
    model = model.to(DEV).train(g.meta["training"])
    perturb = g.inputs.get("perturb")
    if perturb is not None:
        perturb = perturb.to(DEV).requires_grad_(True)
    out = model(g.batch().to(DEV), perturb)
    (out * g.inputs["w0"].to(DEV)).sum().backward()
    _close64(out.detach(), g, "out", "0", "out")
    if perturb is not None:
        _close64(perturb.grad, g, "gin", "perturb", "d perturb")
    for k, p in model.named_parameters():
        if k in g.gsd64:
            _close64(p.grad if p.grad is not None else torch.zeros_like(p), g, "gsd", k, f"grad {k}")


@pytest.mark.parametrize("name", golden_names("G12_pnatrans"))
def test_pna_transformer_golden(name):
    from graphtrans_amd.models.pna_transformer import PNATransformer
    from helpers import load_sd, node_encoder

    g = Golden(name)
    args = _golden_args(g)
    model = PNATransformer(g.meta["num_tasks"], node_encoder("code2", args.gnn_emb_dim), None, args)
    load_sd(model, g.sd)
    model = model.to(DEV).train(g.meta["training"])
    out = model(g.batch().to(DEV))
    out = out if isinstance(out, list) else [out]
    sum((o * g.inputs[f"w{i}"].to(DEV)).sum() for i, o in enumerate(out)).backward()
    for i, o in enumerate(out):
        _close64(o.detach(), g, "out", str(i), f"out{i}")
    for k, p in model.named_parameters():
        if k in g.gsd64:
            _close64(p.grad if p.grad is not None else torch.zeros_like(p), g, "gsd", k, f"grad {k}")


def test_pna_transformer_c4_shape_vs_oracle():
    """BASELINE config 4 at its real shape: Code2-like graphs, 128 per batch, D = 272, L = 4, towers 4,
    mean/max/min/std x identity/amplification/attenuation, d_model 128 x 4 encoder layers, 5 heads x 5002
    (configs/code2/pna-transformer/pooling=cls+norm_input.yml:16-23) against the float64 oracle."""
    from graphtrans_amd import losses, synth
    from graphtrans_amd.encoders import ASTNodeEncoder
    from graphtrans_amd.models.pna_transformer import PNATransformer
    from oracle import reference_math as rm

    torch.manual_seed(3)
    args = rm.default_args(gnn_virtual_node=False, gnn_num_layer=4, gnn_emb_dim=272, gnn_JK="last", gnn_residual=True,
                           gnn_dropout=0.0, d_model=128, nhead=4, dim_feedforward=512, transformer_dropout=0.0,
                           num_encoder_layers=4, transformer_norm_input=True, graph_pooling="cls", max_seq_len=5,
                           aggregators=["mean", "max", "min", "std"], scalers=["identity", "amplification", "attenuation"],
                           deg=torch.tensor([0, 4000, 2500, 900, 300, 90, 30, 9]))
    b = synth.code2_like(B=128, seed=5, mean_nodes=40.0, max_nodes=150)   # C4's batch of 128 graphs, sized for the CPU oracle
    model = PNATransformer(5002, ASTNodeEncoder(272, 98, 10030, 20), None, args).train()
    sd = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    torch.set_default_dtype(torch.float64)
    try:
        ref = rm.pna_transformer(sd, args, b, None, True)
        ref_loss = rm.code2_loss(ref, b.y_arr)
        ref_loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    model = model.to(DEV)
    bd = b.to(DEV)
    out = model(bd)
    loss = losses.code2_loss(out, bd.y_arr)
    loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * max(1.0, abs(float(ref_loss)))
    for i, (o, r) in enumerate(zip(out, ref)):
        assert_close(o.detach().cpu(), r.detach(), what=f"out{i}")
    for k, p in model.named_parameters():
        if sd[k].grad is not None:
            assert_close(p.grad.cpu(), sd[k].grad, what=f"grad {k}")


@pytest.mark.parametrize("dropout,aggs,scalers", [(0.0, ["mean", "max", "min", "std"], ["identity", "amplification", "attenuation"]),
                                                   (0.0, ["mean", "std"], ["amplification", "attenuation"]),
                                                   (0.25, ["mean", "max", "min", "std"], ["identity", "amplification", "attenuation"])])
def test_pna_fused_path_equals_module_path(dropout, aggs, scalers):
    """PNATransformer on the fused path (engine.py -> gt_model_forward / gt_model_backward with gt_pna_layer_*) against the
    module-by-module path: same parameters, same batch; outputs and every parameter gradient agree to GEMM-reordering noise.
    With GNN dropout > 0 the two paths draw different masks, so only eligibility, finiteness and the eval-mode outputs are compared."""
    import copy

    from graphtrans_amd import engine, losses, synth
    from graphtrans_amd.encoders import ASTNodeEncoder
    from graphtrans_amd.models.pna_transformer import PNATransformer
    from oracle import reference_math as rm

    torch.manual_seed(11)
    args = rm.default_args(gnn_virtual_node=False, gnn_num_layer=3, gnn_emb_dim=64, gnn_JK="last", gnn_residual=True,
                           gnn_dropout=dropout, d_model=64, nhead=4, dim_feedforward=128, transformer_dropout=0.0,
                           num_encoder_layers=2, transformer_norm_input=True, graph_pooling="cls", max_seq_len=3,
                           aggregators=aggs, scalers=scalers, deg=torch.tensor([0, 40, 25, 9, 3]))
    b = synth.code2_like(B=24, seed=2, mean_nodes=20.0, max_nodes=60).to(DEV)
    model = PNATransformer(50, ASTNodeEncoder(64, 98, 10030, 20), None, args).to(DEV).train()
    ref = copy.deepcopy(model)
    ref.fused = False
    assert engine.eligible(model, b, None) and not engine.eligible(ref, b, None)
    model.eval(), ref.eval()   # (before any training step: the running statistics are still identical)
    with torch.no_grad():
        assert engine.eligible(model, b, None)
        for o, r in zip(model(b), ref(b)):
            assert_close(o.cpu(), r.cpu(), what="eval out")
    model.train(), ref.train()
    if dropout == 0.0:
        for m in (model, ref):
            loss = losses.code2_loss(m(b), b.y_arr)
            loss.backward()
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert p.grad is not None and q.grad is not None, k
            assert_close(p.grad.cpu(), q.grad.cpu(), what=f"grad {k}")
    else:
        loss = losses.code2_loss(model(b), b.y_arr)
        loss.backward()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_pna_transformer_unaligned_tower_width_vs_oracle():
    """gnn_emb_dim 300 with 4 towers (the reference's default width, modules/pna/pna_module.py:43-51): F = 75 is not a multiple of 4;
    the conv runs on zero-padded tower rows (module path) and must match the float64 oracle like the aligned widths do."""
    from graphtrans_amd import engine
    from graphtrans_amd.encoders import ASTNodeEncoder
    from graphtrans_amd.models.pna_transformer import PNATransformer
    from oracle import reference_math as rm

    torch.manual_seed(4)
    args = rm.default_args(gnn_virtual_node=False, gnn_num_layer=2, gnn_emb_dim=300, gnn_JK="last", gnn_residual=True,
                           gnn_dropout=0.0, d_model=32, nhead=2, dim_feedforward=48, transformer_dropout=0.0,
                           num_encoder_layers=1, transformer_norm_input=True, graph_pooling="cls", max_seq_len=None,
                           aggregators=["mean", "max", "min", "std"], scalers=["identity", "amplification", "attenuation"],
                           deg=torch.tensor([0, 5, 9, 7, 3, 1, 0, 2]))
    b = _graph(9, with_dupes=False)
    model = PNATransformer(7, ASTNodeEncoder(300, 11, 13, 20), None, args)
    _randomize(model, 5)
    model.train()
    sd = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    torch.set_default_dtype(torch.float64)
    try:
        ref = rm.pna_transformer(sd, args, b, None, True)
        w = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
        (ref * w).sum().backward()
    finally:
        torch.set_default_dtype(torch.float32)
    model = model.to(DEV)
    bd = b.to(DEV)
    assert not engine.eligible(model, bd, None)   # the fused layer wants 16-byte tower rows
    out = model(bd)
    (out * w.float().to(DEV)).sum().backward()
    assert_close(out.detach().cpu(), ref.detach(), what="out")
    for k, p in model.named_parameters():
        if sd[k].grad is not None:
            assert_close(p.grad.cpu(), sd[k].grad, what=f"grad {k}")
