"""Input encoders: gt_embed_sum_fwd / _bwd against torch.nn.Embedding sums
(reference: ASTNodeEncoder.forward dataset/utils.py:28-30, ogb AtomEncoder dataset/mol.py:83)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(cols, tables, clamps):
    out = 0
    for c, t, k in zip(cols, tables, clamps):
        if k is not None:
            c = c.clamp(max=k)
        out = out + t[c]
    return out


@pytest.mark.parametrize("N,D,rows,clamps", [
    (1000, 128, [98, 1003, 21], [None, None, 20]),
    (5000, 300, [98, 10030, 21], [None, None, 20]),
    (777, 64, [119, 4, 12, 12, 10, 6, 6, 2, 2], [None] * 9),
    (1, 32, [5], [None]),
])
def test_embed_sum_fwd_bwd(N, D, rows, clamps):
    from graphtrans_amd import ops
    g = torch.Generator().manual_seed(N + D)
    x = torch.stack([torch.randint(0, r, (N,), generator=g) for r in rows], 1)
    if clamps[-1] is not None:  # depth column may exceed max_depth before the clamp
        x[:, -1] = torch.randint(0, 3 * rows[-1], (N,), generator=g)
    x = x.cuda()
    tables = [torch.randn(r, D, generator=g).cuda().requires_grad_() for r in rows]
    cols = [x[:, i] for i in range(len(rows))]
    out = ops.embed_sum(cols, tables, clamps)
    ref = _ref(cols, [t.detach() for t in tables], clamps)
    assert torch.equal(out, ref)  # same summation order -> bit-exact

    gout = (torch.randn(N, D, generator=g) * torch.logspace(-3, 1, N).unsqueeze(1)).cuda()
    out.backward(gout)
    got = [t.grad.clone() for t in tables]
    t64 = [t.detach().double().requires_grad_() for t in tables]
    _ref(cols, t64, clamps).backward(gout.double())
    scale = gout.abs().max().item()
    for a, b, r in zip(got, t64, rows):
        # fixed point: each addend is rounded to 2^-30 of max|g| (<= 2^-29 * max after the pow2 scale)
        n_max = max(int(N), 1)
        assert (a.double() - b.grad).abs().max().item() <= scale * 2.0 ** -29 * n_max + 1e-6 * b.grad.abs().max().item()
    # deterministic: a second run is bit-identical
    for t in tables:
        t.grad = None
    ops.embed_sum(cols, tables, clamps).backward(gout)
    for a, t in zip(got, tables):
        assert torch.equal(a, t.grad)


def test_embed_sum_empty_and_zero_grad():
    from graphtrans_amd import ops
    tab = torch.randn(7, 16).cuda().requires_grad_()
    idx = torch.zeros(0, dtype=torch.int64).cuda()
    out = ops.embed_sum([idx], [tab])
    assert out.shape == (0, 16)
    idx = torch.arange(5).cuda()
    out = ops.embed_sum([idx], [tab])
    out.backward(torch.zeros_like(out))
    assert torch.equal(tab.grad, torch.zeros_like(tab))


def test_ast_node_encoder_matches_reference_formula():
    from graphtrans_amd.encoders import ASTNodeEncoder
    torch.manual_seed(0)
    enc = ASTNodeEncoder(64, 11, 13, 20).cuda()
    x = torch.stack([torch.randint(0, 11, (300,)), torch.randint(0, 13, (300,))], 1).cuda()
    depth = torch.randint(0, 50, (300, 1)).cuda()
    out = enc(x, depth.view(-1))
    d = depth.view(-1).clamp(max=20)
    ref = enc.type_encoder.weight[x[:, 0]] + enc.attribute_encoder.weight[x[:, 1]] + enc.depth_encoder.weight[d]
    assert torch.equal(out, ref)
    assert depth.max().item() > 20  # caller's tensor is not clamped in place


def test_embed_sum_bwd_hot_row():
    """A skewed index column (most nodes share one row of a large table): same result, and the
    register run-length aggregation keeps it from serialising on that row."""
    from graphtrans_amd import ops
    g = torch.Generator().manual_seed(1)
    N, D, R = 20000, 128, 5000
    idx = torch.randint(0, R, (N,), generator=g)
    idx[torch.rand(N, generator=g) < 0.8] = 17
    idx = idx.cuda()
    tab = torch.randn(R, D, generator=g).cuda().requires_grad_()
    out = ops.embed_sum([idx], [tab])
    gout = torch.randn(N, D, generator=g).cuda()
    out.backward(gout)
    ref = torch.zeros(R, D, dtype=torch.float64, device="cuda").index_add_(0, idx, gout.double())
    scale = gout.abs().max().item()
    assert (tab.grad.double() - ref).abs().max().item() <= scale * 2.0 ** -29 * N + 1e-6 * ref.abs().max().item()


@pytest.mark.parametrize("case", ["code2", "atom", "hot", "one_row", "tiny", "chunk_edges"])
def test_embed_sorted_backward(case):
    """gt_embed_sort + gt_embed_sum_bwd_sorted: every table row is the fp32 sum of its nodes' gradient rows in node
    order, associated per 64-position chunk -> equals the float64 sum within fp32 rounding of the partial sums."""
    from graphtrans_amd import ops
    g = torch.Generator().manual_seed(5)
    if case == "code2":
        N, D, rows, clamps = 3000, 300, [98, 10030, 21], [None, None, 20]
    elif case == "atom":
        N, D, rows, clamps = 700, 64, [119, 4, 12, 12, 10, 6, 6, 2, 2], None
    elif case == "hot":
        N, D, rows, clamps = 5000, 32, [50, 4000], None
    elif case == "one_row":
        N, D, rows, clamps = 1000, 16, [1], None
    elif case == "tiny":
        N, D, rows, clamps = 1, 8, [5, 3], None
    else:   # segments that end exactly on / one off the 64-position chunk boundaries
        N, D, rows, clamps = 64 * 5, 8, [7], None
    idx = [torch.randint(0, r + (5 if clamps and clamps[t] is not None else 0), (N,), generator=g) for t, r in enumerate(rows)]
    if case == "hot":
        idx[1][torch.rand(N, generator=g) < 0.85] = 7
    if case == "chunk_edges":
        idx[0] = torch.tensor([0] * 64 + [1] * 63 + [2] * 1 + [3] * 65 + [4] * 127).long()[torch.arange(N)]
        idx[0] = idx[0][torch.randperm(N, generator=g)]
    tabs = [torch.randn(r, D, generator=g).cuda().requires_grad_() for r in rows]
    cols = [i.cuda() for i in idx]
    out = ops.embed_sum(cols, tabs, clamps=clamps)
    gout = torch.randn(N, D, generator=g)
    out.backward(gout.cuda())
    for t, r in enumerate(rows):
        key = idx[t].clamp(max=clamps[t]) if clamps and clamps[t] is not None else idx[t]
        want = torch.zeros(r, D, dtype=torch.float64).index_add_(0, key, gout.double())
        mag = torch.zeros(r, D, dtype=torch.float64).index_add_(0, key, gout.double().abs())   # sum of |terms| per entry
        got = tabs[t].grad.cpu().double()
        assert ((got - want).abs() <= 4e-7 * mag + 1e-30).all(), (case, t, (got - want).abs().max())
        untouched = torch.bincount(key, minlength=r) == 0
        assert (got[untouched] == 0).all()


def test_embed_sorted_backward_is_deterministic_and_matches_fixed_point_path():
    from graphtrans_amd import ops
    g = torch.Generator().manual_seed(6)
    N, D, rows = 20000, 300, [98, 10030, 21]
    cols = [torch.randint(0, r, (N,), generator=g).cuda() for r in rows]
    gout = torch.randn(N, D, generator=g).cuda()
    res = []
    for limit in (16384, 16384, 0):    # 0: force the fixed-point atomic path
        old, ops.EMBED_SORT_MAX_ROWS = ops.EMBED_SORT_MAX_ROWS, limit
        try:
            tabs = [torch.zeros(r, D, device="cuda").requires_grad_() for r in rows]
            ops.embed_sum(cols, tabs).backward(gout)
            res.append([t.grad.clone() for t in tabs])
        finally:
            ops.EMBED_SORT_MAX_ROWS = old
    for a, b, c in zip(*res):
        assert torch.equal(a, b)
        assert (a - c).abs().max().item() <= 1e-5 * max(1.0, a.abs().max().item())
