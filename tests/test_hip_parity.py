"""GPU parity: graphtrans_amd (HIP kernels through the C ABI) vs the golden vectors generated
from the reference and vs the CPU oracle on seeded inputs.

fp32 tolerance 1e-4 (scale-relative, see conftest.assert_close) = BASELINE.json north_star;
index structures bit-exact; bf16 storage mode 3e-2 (bf16 has 8 mantissa bits; stated per test).
"""
import numpy as np
import pytest
import torch

from conftest import Golden, assert_close, golden_names
from helpers import edge_cls, grads_of, load_sd, node_encoder

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _to_dev(b):
    return b.to(DEV)


# ------------------------------------------------------------------------------------------------
# G10 + seeded: graph structure, bit exact
# ------------------------------------------------------------------------------------------------
def _check_struct(edge_index, batch, num_graphs=None):
    from graphtrans_amd.graph import GraphStructure
    from oracle.graph_struct import graph_struct

    gs = GraphStructure.build(edge_index.to(DEV), batch.to(DEV), num_graphs=num_graphs)
    gs.validate()
    ref = graph_struct(edge_index.numpy(), batch.numpy(), num_graphs=num_graphs)
    E = edge_index.shape[1]
    got = dict(ptr=gs.graph_ptr, in_ptr=gs.in_ptr, in_eid=gs.in_eid[:E], in_src=gs.in_src[:E], out_ptr=gs.out_ptr,
               out_eid=gs.out_eid[:E], out_dst=gs.out_dst[:E])
    for k, v in got.items():
        assert np.array_equal(v.cpu().numpy().astype(np.int64), ref[k]), k
    n = batch.numel()
    assert np.array_equal(gs.deg[:n].cpu().numpy(), (ref["deg_out"] + 1).astype(np.float32))
    assert np.allclose(gs.dis[:n].cpu().numpy(), (ref["deg_out"] + 1.0) ** -0.5, rtol=1e-6)
    assert np.array_equal(gs.node_graph[:n].cpu().numpy(), batch.numpy().astype(np.int32))
    return gs


@pytest.mark.parametrize("name", golden_names("G10_"))
def test_graph_struct_golden(name):
    g = Golden(name)
    _check_struct(g.inputs["edge_index"], g.inputs["batch"])


def test_graph_struct_seeded_and_hubs():
    from graphtrans_amd import synth

    b = synth.code2_like(B=64, seed=5)
    _check_struct(b.edge_index, b.batch, num_graphs=64)
    # hub rows exercise the bitonic (33..4096) and the rank-by-counting (>4096) sort tiers
    rng = np.random.default_rng(0)
    n = 6000
    src = np.concatenate([rng.integers(0, n, 5000), np.full(3000, 7), rng.integers(0, n, 300)])
    dst = np.concatenate([np.full(5000, 3), rng.integers(0, n, 3000), np.full(300, 11)])
    perm = rng.permutation(src.size)
    ei = torch.from_numpy(np.stack([src[perm], dst[perm]]).astype(np.int64))
    batch = torch.zeros(n, dtype=torch.int64)
    _check_struct(ei, batch, num_graphs=1)
    # no edges at all; graphs with a single node; a trailing empty graph id
    _check_struct(torch.zeros((2, 0), dtype=torch.int64), torch.tensor([0, 1, 1, 3]), num_graphs=5)


def test_graph_struct_rejects_bad_index():
    from graphtrans_amd.graph import GraphStructure

    ei = torch.tensor([[0, 9], [1, 0]], device=DEV)
    gs = GraphStructure.build(ei, torch.zeros(3, dtype=torch.int64, device=DEV), num_graphs=1)
    with pytest.raises(ValueError):
        gs.validate()


# ------------------------------------------------------------------------------------------------
# G1 / G2 convolutions
# ------------------------------------------------------------------------------------------------
def _run_and_compare(g, module, fwd, float_inputs, atol=1e-4):
    outs = fwd()
    outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
    assert len(outs) == len(g.out_list)
    loss = 0
    for i, o in enumerate(outs):
        assert_close(o.detach().cpu(), g.out_list[i], atol=atol, rtol=atol, what=f"{g.name} out{i}")
        loss = loss + (o * g.inputs[f"w{i}"].to(DEV)).sum()
    if not g.gsd and not g.gin:
        return
    loss.backward()
    got = grads_of(module)
    for k, v in g.gsd.items():
        assert_close(got[k].cpu(), v, atol=atol, rtol=atol, what=f"{g.name} grad {k}")
    for k, v in g.gin.items():
        assert_close(float_inputs[k].grad.cpu(), v, atol=atol, rtol=atol, what=f"{g.name} grad input {k}")


@pytest.mark.parametrize("name", golden_names("G1_") + golden_names("G2_"))
def test_conv_golden(name):
    from graphtrans_amd.modules.conv import GCNConv, GINConv

    g = Golden(name)
    cls = GCNConv if g.meta["conv"] == "gcn" else GINConv
    conv = load_sd(cls(g.meta["D"], edge_cls(g.meta["edge"])), g.sd).to(DEV)
    conv.train(g.meta["training"])
    x = g.inputs["x"].to(DEV).requires_grad_(True)
    ei = g.inputs["edge_index"].to(DEV)
    ea = g.inputs["edge_attr"].to(DEV) if "edge_attr" in g.inputs else None
    _run_and_compare(g, conv, lambda: conv(x, ei, ea), {"x": x})


@pytest.mark.parametrize("conv_name,edge,D,dtype", [
    ("gcn", "linear", 300, torch.float32), ("gin", "linear", 300, torch.float32), ("gcn", "none", 128, torch.float32),
    ("gin", "bond", 300, torch.float32), ("gcn", "dense", 272, torch.float32), ("gcn", "linear", 1024, torch.float32),
    ("gcn", "linear", 300, torch.bfloat16), ("gin", "bond", 128, torch.bfloat16),
    ("gcn", "linear", 256, torch.float32), ("gcn", "linear", 256, torch.bfloat16),   # D = 256: the ER stress config (C5)
    # the W-floats-per-lane kernels (aggregate_wide.h): W = 5 (300, 320), 6 (384), 7 (448) x every edge mode / conv
    ("gcn", "dense", 300, torch.float32), ("gin", "none", 300, torch.float32), ("gcn", "bond", 320, torch.float32),
    ("gin", "linear", 384, torch.float32), ("gcn", "bond", 384, torch.float32), ("gin", "dense", 448, torch.float32),
    ("gcn", "linear", 448, torch.float32), ("gcn", "none", 360, torch.float32)])
def test_aggregate_vs_oracle(conv_name, edge, D, dtype):
    """Seeded Code2 / Molpcba-shaped batches at the real emb dims; checked against the CPU oracle
    (gcn_aggregate / gin_aggregate) incl. every parameter gradient."""
    from graphtrans_amd import ops, synth
    from graphtrans_amd.graph import GraphStructure
    from graphtrans_amd.modules.conv import edge_spec
    from oracle import reference_math as rm

    torch.manual_seed(0)
    if D == 256:   # BASELINE configs[4]: G(512, 8/511) graphs, both edge directions stored
        b = synth.er_stress(B=3, seed=3)
    else:
        b = synth.molpcba_like(B=24, seed=3) if edge == "bond" else synth.code2_like(B=12, seed=3)
    N = b.num_nodes
    h = torch.randn(N, D).to(dtype).float()  # the oracle sees the same (storage-rounded) inputs
    w = torch.randn(N, D)
    enc = None
    if edge == "linear":
        enc = torch.nn.Linear(2, D)
    elif edge == "bond":
        from graphtrans_amd.encoders import BondEncoder
        enc = BondEncoder(D)
    elif edge == "dense":
        enc = torch.nn.Sequential(torch.nn.Linear(2, 8), torch.nn.Tanh(), torch.nn.Linear(8, D))
    self_param = torch.randn(1, D) * 0.3 if conv_name == "gcn" else torch.tensor([0.3])
    # ---- oracle (fp32 CPU)
    h_ref = h.clone().requires_grad_(True)
    sp_ref = self_param.clone().requires_grad_(True)
    if edge == "bond":  # the encoder module itself is GPU-only; restate ogb's BondEncoder sum for the oracle
        e_ref = sum(emb(b.edge_attr[:, i]) for i, emb in enumerate(enc.bond_embedding_list))
    else:
        e_ref = enc(b.edge_attr) if enc is not None else None
    if conv_name == "gcn":
        out_ref = rm.gcn_aggregate(h_ref, e_ref, b.edge_index, sp_ref)
    else:
        out_ref = (1 + sp_ref) * h_ref + rm.gin_aggregate(h_ref, e_ref, b.edge_index)
    (out_ref * w).sum().backward()
    ref_pg = {k: p.grad.clone() for k, p in enc.named_parameters()} if enc is not None else {}
    if enc is not None:
        enc.zero_grad()
    # ---- HIP
    gs = GraphStructure.build(b.edge_index.to(DEV), b.batch.to(DEV), num_graphs=b.num_graphs)
    enc_d = enc.to(DEV) if enc is not None else (lambda _: 0)
    h_d = h.to(DEV).to(dtype).requires_grad_(True)
    sp_d = self_param.to(DEV).requires_grad_(True)
    ea_d = b.edge_attr.to(DEV) if b.edge_attr is not None else None
    spec = edge_spec(enc_d, ea_d, D)
    assert spec.kind == {"linear": "linear", "bond": "tables", "dense": "dense", "none": "none"}[edge]
    if spec.kind == "dense" and dtype != torch.float32:
        spec.dense = spec.dense.to(dtype)
    out = ops.aggregate(h_d, gs, conv_name, sp_d, spec)
    (out.float() * w.to(DEV)).sum().backward()
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert_close(out.float().cpu(), out_ref.detach(), atol=tol, rtol=tol, what="out")
    assert_close(h_d.grad.float().cpu(), h_ref.grad, atol=tol, rtol=tol, what="dh")
    assert_close(sp_d.grad.cpu(), sp_ref.grad, atol=tol, rtol=tol, what="d self_param")
    if enc is not None:
        for k, p in enc_d.named_parameters():
            assert_close(p.grad.cpu(), ref_pg[k], atol=tol, rtol=tol, what=f"d {k}")


@pytest.mark.parametrize("conv_name", ["gcn", "gin"])
@pytest.mark.parametrize("D", [128, 300])
def test_aggregate_without_edges(conv_name, D):
    """A batch of isolated nodes (E = 0): only the self term is left, forward and backward."""
    from graphtrans_amd import ops
    from graphtrans_amd.graph import GraphStructure

    torch.manual_seed(2)
    N = 37
    ei = torch.zeros(2, 0, dtype=torch.long, device=DEV)
    batch = torch.arange(N, device=DEV) // 5
    gs = GraphStructure.build(ei, batch, num_graphs=int(batch.max()) + 1)
    h = torch.randn(N, D, device=DEV, requires_grad=True)
    sp = (torch.randn(1, D, device=DEV) * 0.3 if conv_name == "gcn" else torch.tensor([0.3], device=DEV)).requires_grad_(True)
    lin = torch.nn.Linear(2, D).to(DEV)
    spec = ops.EdgeSpec("linear", attr=torch.zeros(0, 2, device=DEV), weight=lin.weight, bias=lin.bias)
    w = torch.randn(N, D, device=DEV)
    out = ops.aggregate(h, gs, conv_name, sp, spec)
    (out * w).sum().backward()
    h_ref = h.detach().clone().requires_grad_(True)
    sp_ref = sp.detach().clone().requires_grad_(True)
    ref = torch.relu(h_ref + sp_ref) if conv_name == "gcn" else (1 + sp_ref) * h_ref   # deg = 1 (the self loop), conv.py:63-65
    (ref * w).sum().backward()
    assert_close(out.detach().cpu(), ref.detach().cpu(), what="out")
    assert_close(h.grad.cpu(), h_ref.grad.cpu(), what="dh")
    assert_close(sp.grad.cpu(), sp_ref.grad.cpu(), what="d self_param")


def test_aggregate_is_deterministic_and_edge_order_invariant():
    """CSR summation order is fixed (bitwise reproducible) and a permutation of the edge list
    changes the result only by fp32 reassociation."""
    from graphtrans_amd import ops, synth
    from graphtrans_amd.graph import GraphStructure

    torch.manual_seed(1)
    b = synth.code2_like(B=16, seed=9).to(DEV)
    D = 300
    h = torch.randn(b.num_nodes, D, device=DEV)
    lin = torch.nn.Linear(2, D).to(DEV)
    root = torch.randn(1, D, device=DEV)

    def run(ei, ea):
        gs = GraphStructure.build(ei, b.batch, num_graphs=16)
        return ops.aggregate(h, gs, "gcn", root, ops.EdgeSpec("linear", attr=ea, weight=lin.weight, bias=lin.bias))

    o1, o2 = run(b.edge_index, b.edge_attr), run(b.edge_index, b.edge_attr)
    assert torch.equal(o1, o2)
    perm = torch.randperm(b.edge_index.shape[1], device=DEV)
    o3 = run(b.edge_index[:, perm].contiguous(), b.edge_attr[perm].contiguous())
    assert_close(o3.cpu(), o1.cpu(), what="edge permutation")


# ------------------------------------------------------------------------------------------------
# G3 / G4 GNN stacks, G5 pad, G6 encoder, G8 model
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names("G3_") + golden_names("G4_"))
def test_gnn_node_golden(name):
    from graphtrans_amd.modules.gnn_module import GNNNodeEmbedding

    g = Golden(name)
    a = g.args()
    m = GNNNodeEmbedding(a.gnn_virtual_node, a.gnn_num_layer, a.gnn_emb_dim, node_encoder(g.meta["feat"], a.gnn_emb_dim),
                         edge_cls(g.meta["edge"]), JK=a.gnn_JK, drop_ratio=a.gnn_dropout, residual=a.gnn_residual,
                         gnn_type=a.gnn_type)
    m = load_sd(m, g.sd).to(DEV)
    m.train(g.meta["training"])
    b = _to_dev(g.batch())
    fi = {}
    if b.x.is_floating_point():
        b.x = b.x.requires_grad_(True)
        fi["x"] = b.x
    perturb = g.inputs.get("perturb")
    if perturb is not None:
        perturb = perturb.to(DEV).requires_grad_(True)
        fi["perturb"] = perturb
    _run_and_compare(g, m, lambda: m(b, perturb), fi)


@pytest.mark.parametrize("name", golden_names("G5_"))
def test_pad_golden(name):
    from graphtrans_amd.modules.utils import pad_batch, unpad_batch

    g = Golden(name)
    h = g.inputs["h"].to(DEV).requires_grad_(True)
    batch = g.inputs["batch"].to(DEV)
    padded, mask, num_nodes, origin, S = pad_batch(h, batch, g.meta["max_input_len"], get_mask=True)
    assert torch.equal(padded.detach().cpu(), g.outs["0"])  # pure copy: bit exact
    assert torch.equal(mask.cpu(), g.outs["1"])
    (padded * g.inputs["w0"].to(DEV)).sum().backward()
    assert torch.equal(h.grad.cpu(), g.gin["h"])
    unp = unpad_batch(g.inputs["padded_in"].to(DEV), g.inputs["prev"].to(DEV), num_nodes, origin, S)
    assert torch.equal(unp.cpu(), g.outs["2"])
    # the reference's five return values (modules/utils.py:27-29): masks[i] == batch.eq(i), num_nodes[i] its count,
    # S = min(max nodes, max_input_len); unpad_batch also takes the masks as a plain list of tensors
    B = int(batch[-1]) + 1
    assert len(origin) == B and len(num_nodes) == B
    for i in (0, B - 1):
        assert torch.equal(origin[i], batch.eq(i)) and int(num_nodes[i]) == int(batch.eq(i).sum())
    assert S == min(int(max(num_nodes)), g.meta["max_input_len"])
    unp2 = unpad_batch(g.inputs["padded_in"].to(DEV), g.inputs["prev"].to(DEV), num_nodes, [origin[i] for i in range(B)], S)
    assert torch.equal(unp2.cpu(), g.outs["2"])


@pytest.mark.parametrize("name", golden_names("G6_"))
def test_transformer_node_encoder_golden(name):
    from graphtrans_amd.modules.transformer_encoder import TransformerNodeEncoder

    g = Golden(name)
    m = load_sd(TransformerNodeEncoder(g.args()), g.sd).to(DEV)
    m.train(g.meta["training"])
    x = g.inputs["padded"].to(DEV).requires_grad_(True)
    mask = g.inputs["mask"].to(DEV)
    _run_and_compare(g, m, lambda: m(x, mask)[0], {"padded": x})


@pytest.mark.parametrize("name", golden_names("G7_"))
def test_masked_encoder_golden(name):
    from graphtrans_amd.modules.masked_transformer_encoder import Block, CausalSelfAttention

    g = Golden(name)
    d = g.inputs["x"].shape[-1]
    if g.meta["kind"] == "causal_self_attention":
        m = CausalSelfAttention(d, g.meta["n_head"], 0.0, 0.0)
    else:
        m = Block(d, g.sd["mlp.0.weight"].shape[0], g.meta["n_head"], 0.0, 0.0, prenorm=g.meta["prenorm"])
    m = load_sd(m, g.sd).to(DEV)
    x = g.inputs["x"].to(DEV).requires_grad_(True)
    adj = g.inputs["attn_mask"].to(DEV) if "attn_mask" in g.inputs else None
    valid = g.inputs["valid_input_mask"].to(DEV) if "valid_input_mask" in g.inputs else None
    _run_and_compare(g, m, lambda: m(x, adj, valid), {"x": x})


@pytest.mark.parametrize("name", golden_names("G8_"))
@pytest.mark.parametrize("layout", ["auto", "padded"])
def test_gnn_transformer_golden(name, layout):
    from graphtrans_amd.models.gnn_transformer import GNNTransformer

    g = Golden(name)
    a = g.args()
    a.token_layout = layout
    m = GNNTransformer(g.meta["num_tasks"], node_encoder(g.meta["feat"], a.gnn_emb_dim), edge_cls(g.meta["edge"]), a)
    m = load_sd(m, g.sd).to(DEV)
    m.train(g.meta["training"])
    b = _to_dev(g.batch())
    fi = {}
    if b.x.is_floating_point():
        b.x = b.x.requires_grad_(True)
        fi["x"] = b.x
    _run_and_compare(g, m, lambda: m(b), fi)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ["code2", "one_giant", "empties", "small"])
def test_segment_sum_two_phase(case, dtype):
    """global_add_pool(h, batch) + vn (modules/gnn_module.py:219): the load-balanced two-phase kernel (row chunks +
    per-graph fix-up, N >= 4096) and the one-block-per-graph kernel against a float64 index_add, with graphs that
    sit inside one chunk, end exactly on chunk boundaries, span many chunks, or are empty."""
    from graphtrans_amd import ops, synth
    from graphtrans_amd.graph import GraphStructure
    g = torch.Generator().manual_seed(3)
    if case == "code2":
        sizes = torch.bincount(synth.code2_like(B=256, seed=0).batch, minlength=256)
    elif case == "one_giant":
        sizes = torch.tensor([64, 1, 63, 128, 9000, 2, 64, 65])
    elif case == "empties":
        sizes = torch.tensor([0, 0, 70, 0, 5000, 0, 0, 58, 6, 0])
    else:
        sizes = torch.tensor([5, 0, 17, 1])
    B, N, D = sizes.numel(), int(sizes.sum()), 300
    batch = torch.repeat_interleave(torch.arange(B), sizes)
    gs = GraphStructure.build(torch.zeros(2, 0, dtype=torch.int64, device=DEV), batch.to(DEV), num_graphs=B)
    x = torch.randn(N, D, generator=g).to(dtype)
    add = torch.randn(B, D, generator=g).to(dtype)
    got = ops._segment_sum_raw(x.to(DEV), add.to(DEV), gs).float().cpu()
    want = add.double().index_add(0, batch, x.double())
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    scale = max(1.0, want.abs().max().item())
    assert (got.double() - want).abs().max().item() <= tol * scale
    got0 = ops._segment_sum_raw(x.to(DEV), None, gs).float().cpu()
    want0 = torch.zeros(B, D, dtype=torch.float64).index_add(0, batch, x.double())
    assert (got0.double() - want0).abs().max().item() <= tol * scale
    assert torch.equal(got, ops._segment_sum_raw(x.to(DEV), add.to(DEV), gs).float().cpu())   # reproducible
