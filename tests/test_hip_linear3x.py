"""bf16x6 GEMMs (csrc/linear3x.h: fp32 operands split exactly into three bf16 planes, six bf16-MFMA products) against a float64
evaluation, with torch's fp32 GEMM and the exact-fp32 MFMA kernels (linear32.h) as yardsticks.  Through the C ABI
(gt_w3_images / gt_w3_bind + gt_linear_fwd_ld2 / gt_linear_bwd_ld2) at the shapes of the benchmarked configurations:
Code2 (31.6 k x 300 x 300, gnn2transformer 600 -> 128), Molpcba GIN (6.7 k x 300 <-> 600), ER (256 x 256), PNA (272)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GT_F32, GT_BF16 = 0, 1


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def _p(t):
    return None if t is None else t.data_ptr()


def fwd(x, W, b, bound, act=0, p=0.0, seed=0, out_dtype=torch.float32):
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty(M, N, dtype=out_dtype, device=x.device)
    code = lambda t: GT_BF16 if t.dtype == torch.bfloat16 else GT_F32
    call = lambda: _lib.launch("gt_linear_fwd_ld2", code(x), code(y), GT_F32, _p(x), _p(W), _p(b), _p(y), M, N, K, K, N, act, p, seed, _stream())
    if bound is not None:
        with bound.bound():
            call()
    else:
        call()
    return y


def dx_of(x_like, W, dy, ymask, add1, add2, bound, p=0.0, dx_dtype=torch.float32):
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    M, N = dy.shape
    K = W.shape[1]
    dx = torch.empty(M, K, dtype=dx_dtype, device=dy.device)
    code = lambda t: GT_BF16 if t.dtype == torch.bfloat16 else GT_F32
    ws_bytes = _lib.lib().gt_linear_bwd_workspace_bytes(GT_F32, M, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device)
    call = lambda: _lib.launch("gt_linear_bwd_ld2", code(dx), code(dy), GT_F32, None, _p(W), _p(dy), _p(ymask), _p(add1), _p(add2), _p(dx), None, None,
                               M, N, K, K, N, p, _p(ws), ws_bytes, _stream())
    if bound is not None:
        with bound.bound():
            call()
    else:
        call()
    return dx


SHAPES = [(31598, 300, 300), (31598, 128, 600), (6700, 600, 300), (6700, 300, 600), (4100, 256, 256), (16001, 272, 272), (2048, 64, 36), (1024, 20, 132)]


@pytest.mark.parametrize("M,N,K", SHAPES, ids=[f"{m}x{n}x{k}" for m, n, k in SHAPES])
def test_forward_and_dx_match_float64_like_an_fp32_gemm(M, N, K):
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV) * (1.0 + 3.0 * torch.rand(M, 1, device=DEV))
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    imgs = W3Images([W])
    imgs.build()
    y3, y32 = fwd(x, W, b, imgs), fwd(x, W, b, None)
    y64 = torch.nn.functional.linear(x.double(), W.double(), b.double())
    t32 = torch.nn.functional.linear(x, W, b)
    e3, e32, et = rel(y3, y64), rel(y32, y64), rel(t32, y64)
    print(f"\nfwd {M}x{N}x{K}: bf16x6 {e3:.2e}  exact-fp32 MFMA {e32:.2e}  torch fp32 {et:.2e}")
    assert e3 <= max(3 * et, 1e-6), (e3, et)
    assert float((y3 - y64).abs().max()) <= 1e-4 * max(1.0, float(y64.abs().max()))
    # dX = dY W (+ addends), dZ = dY * 1[Y > 0] / keep
    dy = torch.randn(M, N, device=DEV)
    yf = torch.relu(torch.randn(M, N, device=DEV))
    a1, a2 = torch.randn(M, K, device=DEV), torch.randn(M, K, device=DEV)
    d3, d32 = dx_of(x, W, dy, yf, a1, a2, imgs, p=0.25), dx_of(x, W, dy, yf, a1, a2, None, p=0.25)
    dz = (dy.double() * (yf > 0)) / 0.75
    d64 = dz @ W.double() + a1.double() + a2.double()
    dt = ((dy * (yf > 0)) / 0.75) @ W + a1 + a2
    e3, e32, et = rel(d3, d64), rel(d32, d64), rel(dt, d64)
    print(f"dX  {M}x{N}x{K}: bf16x6 {e3:.2e}  exact-fp32 MFMA {e32:.2e}  torch fp32 {et:.2e}")
    assert e3 <= max(3 * et, 1e-6), (e3, et)
    # plain dX (no mask, no addends)
    d3 = dx_of(x, W, dy, None, None, None, imgs)
    assert rel(d3, dy.double() @ W.double()) <= max(3 * rel(dy @ W, dy.double() @ W.double()), 1e-6)


def test_epilogue_relu_dropout_is_the_exact_kernels_mask():
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(1)
    M, N, K = 5000, 300, 300
    x, W, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / K ** 0.5, torch.randn(N, device=DEV)
    imgs = W3Images([W])
    imgs.build()
    y3 = fwd(x, W, b, imgs, act=1, p=0.3, seed=12345)
    y32 = fwd(x, W, b, None, act=1, p=0.3, seed=12345)
    # same counter hash -> the same elements are dropped; ReLU gates may differ only where the pre-activation is ~0
    differ = (y3 == 0) != (y32 == 0)
    assert int(differ.sum()) <= 20, int(differ.sum())
    same = ~differ
    assert float((y3[same] - y32[same]).abs().max()) <= 2e-5
    keep = float((y3 != 0).float().mean())
    assert 0.30 < keep < 0.40   # ~ 0.5 (relu) * 0.7 (keep)


def test_bf16_token_rows():
    """gnn2transformer in the mixed mode (models/gnn_transformer.py:69-70,92): fp32 node rows -> bf16 token rows (forward), bf16 d tokens
    -> fp32 d rows (dX: the row operand is bf16, i.e. its own first plane -- three products instead of six)"""
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(2)
    M, N, K = 20000, 128, 600
    x = torch.randn(M, K, device=DEV)
    W, b = torch.randn(N, K, device=DEV) / K ** 0.5, torch.randn(N, device=DEV)
    imgs = W3Images([W])
    imgs.build()
    y3, y32 = fwd(x, W, b, imgs, out_dtype=torch.bfloat16), fwd(x, W, b, None, out_dtype=torch.bfloat16)
    y64 = torch.nn.functional.linear(x.double(), W.double(), b.double())
    assert rel(y3, y64) <= 1.05 * rel(y32, y64) + 1e-6, (rel(y3, y64), rel(y32, y64))
    assert float((y3.float() - y32.float()).abs().max()) <= 2.0 ** -7 * float(y64.abs().max())   # at most one bf16 ulp apart
    dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    d3, d32 = dx_of(x, W, dy, None, None, None, imgs), dx_of(x, W, dy, None, None, None, None)
    d64 = dy.double() @ W.double()
    assert rel(d3, d64) <= max(3 * rel(d32, d64), 1e-6), (rel(d3, d64), rel(d32, d64))
    # with a ReLU gate and keep scale on bf16 rows the operand is no longer bf16: the kernel takes the six-product path
    yf = torch.relu(torch.randn(M, N, device=DEV)).to(torch.bfloat16)
    d3, d32 = dx_of(x, W, dy, yf, None, None, imgs, p=0.2), dx_of(x, W, dy, yf, None, None, None, p=0.2)
    d64 = ((dy.double() * (yf > 0)) / 0.8) @ W.double()
    assert rel(d3, d64) <= max(3 * rel(d32, d64), 1e-6), (rel(d3, d64), rel(d32, d64))


def test_unbound_and_small_calls_keep_the_exact_kernels():
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(3)
    W = torch.randn(300, 300, device=DEV) / 17.0
    other = torch.randn(300, 300, device=DEV) / 17.0
    imgs = W3Images([W])
    imgs.build()
    x = torch.randn(4096, 300, device=DEV)
    a = fwd(x, other, None, imgs)          # a weight that is not in the table
    b = fwd(x, other, None, None)
    assert torch.equal(a, b)
    xs = torch.randn(256, 300, device=DEV)   # short-M GEMMs (the virtual-node MLPs) never take the image path
    assert torch.equal(fwd(xs, W, None, imgs), fwd(xs, W, None, None))


def test_special_values_and_ranges():
    """tiny / huge magnitudes survive the three-way split (bf16 has fp32's exponent range)"""
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(4)
    M, N, K = 2048, 128, 128
    x = torch.randn(M, K, device=DEV) * torch.logspace(-18, 18, M, device=DEV).unsqueeze(1)
    W = torch.randn(N, K, device=DEV)
    imgs = W3Images([W])
    imgs.build()
    y3 = fwd(x, W, None, imgs)
    y64 = x.double() @ W.double().t()
    row_err = (y3.double() - y64).norm(dim=1) / y64.norm(dim=1)
    assert float(row_err.max()) <= 2e-6, float(row_err.max())


def bwd_all(x, W, dy, ymask, bound, p=0.0):
    """dX, dW, db of one gt_linear_bwd_ld2 call"""
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    M, N = dy.shape
    K = W.shape[1]
    dx = torch.empty(M, K, dtype=x.dtype, device=dy.device)
    dw = torch.empty(N, K, device=dy.device)
    db = torch.empty(N, device=dy.device)
    code = lambda t: GT_BF16 if t.dtype == torch.bfloat16 else GT_F32
    ws_bytes = _lib.lib().gt_linear_bwd_workspace_bytes(GT_F32, M, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device)
    call = lambda: _lib.launch("gt_linear_bwd_ld2", code(x), code(dy), GT_F32, _p(x), _p(W), _p(dy), _p(ymask), None, None, _p(dx), _p(dw), _p(db),
                               M, N, K, K, N, p, _p(ws), ws_bytes, _stream())
    if bound is not None:
        with bound.bound():
            call()
    else:
        call()
    return dx, dw, db


DW_SHAPES = [(31598, 300, 300), (31598, 128, 600), (6700, 600, 300), (6700, 300, 600), (4100, 256, 256), (16001, 272, 272), (2048, 64, 36), (1500, 20, 132)]


@pytest.mark.parametrize("M,N,K", DW_SHAPES, ids=[f"{m}x{n}x{k}" for m, n, k in DW_SHAPES])
def test_weight_gradient_matches_float64_like_an_fp32_gemm(M, N, K):
    """k_lin3_dw (both operands split into planes while staged, transposed fragment reads) + the fixed-order reduce"""
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(M * 3 + N + K)
    x = torch.randn(M, K, device=DEV) * (0.5 + torch.rand(M, 1, device=DEV))
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    dy = torch.randn(M, N, device=DEV)
    yf = torch.relu(torch.randn(M, N, device=DEV))
    imgs = W3Images([W])
    imgs.build()
    dz = (dy.double() * (yf > 0)) / 0.9
    w64, b64 = dz.t() @ x.double(), dz.sum(0)
    dzf = (dy * (yf > 0)) / 0.9
    wt, bt = dzf.t() @ x, dzf.sum(0)
    _, w3, b3 = bwd_all(x, W, dy, yf, imgs, p=0.1)
    _, w32, b32 = bwd_all(x, W, dy, yf, None, p=0.1)
    e3, e32, et = rel(w3, w64), rel(w32, w64), rel(wt, w64)
    print(f"\ndW {M}x{N}x{K}: bf16x6 {e3:.2e}  exact-fp32 MFMA {e32:.2e}  torch fp32 {et:.2e};  db {rel(b3, b64):.2e} / {rel(b32, b64):.2e} / {rel(bt, b64):.2e}")
    assert e3 <= max(3 * et, 1e-6), (e3, et)
    assert rel(b3, b64) <= max(3 * rel(bt, b64), 1e-6)
    # bitwise reproducible (fixed-order reduce)
    _, w3b, b3b = bwd_all(x, W, dy, yf, imgs, p=0.1)
    assert torch.equal(w3, w3b) and torch.equal(b3, b3b)


def test_weight_gradient_from_bf16_rows():
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(9)
    M, N, K = 20000, 128, 600
    x = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    imgs = W3Images([W])
    imgs.build()
    _, w3, b3 = bwd_all(x, W, dy, None, imgs)
    w64, b64 = dy.double().t() @ x.double(), dy.double().sum(0)
    assert rel(w3, w64) <= 1e-6 and rel(b3, b64) <= 1e-6, (rel(w3, w64), rel(b3, b64))


def test_virtual_concatenation_equals_the_copied_one():
    """gt_linear_fwd_cat2 / gt_linear_bwd_cat2: [X1 | X2] W^T without building [X1 | X2] (JK = 'cat' feeding gnn2transformer)"""
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(11)
    M, D, N = 31598, 300, 128
    x1, x2 = torch.randn(M, D, device=DEV), torch.randn(M, D, device=DEV)
    W, b = torch.randn(N, 2 * D, device=DEV) / (2 * D) ** 0.5, torch.randn(N, device=DEV)
    cat = torch.cat([x1, x2], 1).contiguous()
    imgs = W3Images([W])
    imgs.build()
    L = _lib.lib()
    for yd in (torch.float32, torch.bfloat16):
        code = GT_BF16 if yd == torch.bfloat16 else GT_F32
        y_ref = fwd(cat, W, b, imgs, out_dtype=yd)
        y = torch.empty(M, N, dtype=yd, device=DEV)
        dy = torch.randn(M, N, device=DEV).to(yd)
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        dw, db = torch.empty_like(W), torch.empty_like(b)
        ws_bytes = L.gt_linear_bwd_workspace_bytes(GT_F32, M, N, 2 * D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        assert L.gt_linear_cat2_ok(GT_F32, _p(W), M, N, D, D) == 0   # nothing bound yet
        with imgs.bound():
            assert L.gt_linear_cat2_ok(GT_F32, _p(W), M, N, D, D) == 1
            _lib.launch("gt_linear_fwd_cat2", code, GT_F32, _p(x1), D, D, _p(x2), D, D, _p(W), _p(b), _p(y), M, N, N, _stream())
            _lib.launch("gt_linear_bwd_cat2", code, GT_F32, _p(x1), D, D, _p(x2), D, D, _p(W), _p(dy), _p(dx1), D, _p(dx2), D, _p(dw), _p(db),
                        M, N, N, _p(ws), ws_bytes, _stream())
        assert torch.equal(y, y_ref)
        dxr, dwr, dbr = bwd_all(cat, W, dy, None, imgs)
        assert torch.equal(torch.cat([dx1, dx2], 1), dxr)
        assert torch.equal(dw, dwr) and torch.equal(db, dbr)


@pytest.mark.parametrize("tok_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("max_len,with_cls", [(1000, 1), (50, 1), (50, 0), (7, 1)])
def test_row_map_equals_pad_and_unpad_passes(tok_dtype, max_len, with_cls):
    """gt_linear_set_rows (gnn2transformer storing the token rows / reading the token-row gradient through gt_seq_token_rows' map,
    models/gnn_transformer.py:92-96) against the same GEMM calls with gt_seq_gather_cls32 behind the forward and gt_seq_scatter in
    front of the backward (modules/utils.py:5-29): bit for bit, truncated graphs (their leading nodes have no token row) included."""
    import numpy as np
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    from graphtrans_amd.w3 import W3Images
    lib = _lib.lib()
    rng = np.random.default_rng(max_len)
    B, K, N = 40, 256, 128
    sizes = rng.integers(1, 120, B)
    sizes[3] = 1
    Nn = int(sizes.sum())
    assert Nn >= 1024
    gptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    kept = np.minimum(sizes, min(int(sizes.max()), max_len))
    kv = kept + with_cls
    tok_ptr = np.concatenate([[0], np.cumsum(kv)])
    desc = np.zeros((B, 4), np.int32)
    desc[:, 0], desc[:, 1], desc[:, 3] = tok_ptr[:-1], kv, kv
    rows = int(tok_ptr[-1])
    last = torch.tensor(tok_ptr[1:] - 1, dtype=torch.int64, device=DEV)
    d_gptr, d_desc = torch.tensor(gptr, device=DEV), torch.tensor(desc, device=DEV)
    d_ng = torch.tensor(np.repeat(np.arange(B, dtype=np.int32), sizes), device=DEV)
    torch.manual_seed(1)
    x = torch.randn(Nn, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    bias = torch.randn(N, device=DEV)
    cls = torch.randn(N, device=DEV)
    tcode = GT_BF16 if tok_dtype == torch.bfloat16 else GT_F32
    imgs = W3Images([W])
    imgs.build()
    st = _stream()
    with imgs.bound():
        assert lib.gt_linear_rows_ok(GT_F32, GT_F32, tcode, _p(W), Nn, N, K) == 1
        # forward
        hn = torch.empty(Nn, N, dtype=tok_dtype, device=DEV)
        _lib.launch("gt_linear_fwd_ld2", GT_F32, tcode, GT_F32, _p(x), _p(W), _p(bias), _p(hn), Nn, N, K, K, N, 0, 0.0, 0, st)
        tok_ref = torch.full((rows, N), 7.0, dtype=tok_dtype, device=DEV)
        _lib.launch("gt_seq_gather_cls32", tcode, _p(hn), _p(cls), _p(d_gptr), _p(d_desc), B, 1, int(kv.max()), with_cls, N, _p(tok_ref), st)
        tok = torch.full((rows, N), 7.0, dtype=tok_dtype, device=DEV)
        rmap = torch.empty(Nn, dtype=torch.int32, device=DEV)
        _lib.launch("gt_seq_token_rows", tcode, _p(cls) if with_cls else None, _p(d_gptr), _p(d_ng), _p(d_desc), B, 1, with_cls, Nn, N, _p(tok), _p(rmap), st)
        _lib.launch("gt_linear_set_rows", _p(rmap))
        _lib.launch("gt_linear_fwd_ld2", GT_F32, tcode, GT_F32, _p(x), _p(W), _p(bias), _p(tok), Nn, N, K, K, N, 0, 0.0, 0, st)
        assert torch.equal(tok, tok_ref)
        want = np.full(Nn, -1, np.int64)
        for b in range(B):
            n0 = gptr[b + 1] - kept[b]
            want[n0:gptr[b + 1]] = tok_ptr[b] + np.arange(kept[b])
        assert np.array_equal(rmap.cpu().numpy().astype(np.int64), want)
        # backward
        dtok = torch.randn(rows, N, device=DEV).to(tok_dtype)
        wsb = lib.gt_linear_bwd_workspace_bytes(GT_F32, Nn, N, K)
        res = []
        for mapped in (False, True):
            ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
            dx, dw, db = torch.empty(Nn, K, device=DEV), torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
            if mapped:
                _lib.launch("gt_linear_set_rows", _p(rmap))
                dy = dtok
            else:
                dy = torch.empty(Nn, N, dtype=tok_dtype, device=DEV)
                _lib.launch("gt_seq_scatter", tcode, _p(dtok), None, _p(d_gptr), _p(d_ng), _p(d_desc), B, 1, with_cls, Nn, N, _p(dy), None, st)
            _lib.launch("gt_linear_bwd_ld2", GT_F32, tcode, GT_F32, _p(x), _p(W), _p(dy), None, None, None, _p(dx), _p(dw), _p(db), Nn, N, K, K, N, 0.0,
                        _p(ws), wsb, st)
            torch.cuda.synchronize()
            res.append((dx, dw, db))
        for a, b_, what in zip(res[0], res[1], ("dx", "dw", "db")):
            assert torch.equal(a, b_), (what, rel(b_, a))
        dropped = torch.tensor(want < 0, device=DEV)
        assert float(res[1][0][dropped].abs().max()) == 0.0 if bool(dropped.any()) else True
        if with_cls:   # the cls gradient: column sums of the CLS rows
            got = torch.empty(N, device=DEV)
            _lib.launch("gt_colsum_rows_f32", tcode, _p(dtok), _p(last), B, N, _p(got), st)
            assert rel(got, dtok[last].double().sum(0).float()) < 1e-5
    # a GEMM that does not take a row map refuses it, and the request does not leak into the next call
    _lib.launch("gt_linear_set_rows", _p(rmap))
    with pytest.raises(RuntimeError):
        _lib.launch("gt_linear_fwd_ld2", GT_F32, tcode, GT_F32, _p(x), _p(W), _p(bias), _p(tok), Nn, N, K, K, N, 0, 0.0, 0, st)   # no bound image
    _lib.launch("gt_linear_fwd_ld2", GT_F32, tcode, GT_F32, _p(x), _p(W), _p(bias), _p(hn), Nn, N, K, K, N, 0, 0.0, 0, st)


def test_deferred_partial_sums_equal_the_immediate_reduces():
    """gt_defer_begin / _flush (csrc/linear.hip): weight-gradient GEMMs and a LayerNorm backward inside a section queue their partial
    sums into ONE launch; dW / db / d gamma / d beta equal the immediate reduces bit for bit (dW: same summation order; the LayerNorm's
    column sums: k_split_reduce's order instead of the finish kernel's -> fp32 order), a full arena falls back to reducing on the
    spot, and an open section without a flush is reported."""
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    from graphtrans_amd.w3 import W3Images
    lib = _lib.lib()
    st = _stream()
    torch.manual_seed(5)
    M, N, K = 9000, 300, 300
    x = torch.randn(M, K, device=DEV)
    dy = torch.randn(M, N, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    imgs = W3Images([W])
    imgs.build()
    wsb = lib.gt_linear_bwd_workspace_bytes(GT_F32, M, N, K)

    def dw_call(bound):
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        dw, db = torch.full((N, K), 7.0, device=DEV), torch.full((N,), 7.0, device=DEV)
        call = lambda: _lib.launch("gt_linear_bwd_ld2", GT_F32, GT_F32, GT_F32, _p(x), _p(W), _p(dy), None, None, None, None, _p(dw), _p(db), M, N, K, K, N,
                                   0.0, _p(ws), wsb, st)
        if bound:
            with imgs.bound():
                call()
        else:
            call()
        return dw, db, ws

    # LayerNorm backward operands
    R, D = 5000, 128
    lx, lg = torch.randn(R, D, device=DEV).bfloat16(), torch.randn(R, D, device=DEV).bfloat16()
    lw = torch.rand(D, device=DEV) + 0.5
    mean, rstd = lx.float().mean(1), 1.0 / (lx.float().var(1, unbiased=False) + 1e-5).sqrt()
    lnb = lib.gt_layernorm_bwd_workspace_bytes(R, D)

    def ln_call():
        ws = torch.empty(lnb, dtype=torch.uint8, device=DEV)
        dx = torch.empty(R, D, dtype=torch.bfloat16, device=DEV)
        gw, gb = torch.full((D,), 7.0, device=DEV), torch.full((D,), 7.0, device=DEV)
        _lib.launch("gt_layernorm_bwd", GT_BF16, _p(lx), None, _p(lg), _p(lw), _p(mean), _p(rstd), 0.0, 0, R, D, _p(dx), None, _p(gw), _p(gb), _p(ws), lnb, st)
        return dx, gw, gb, ws

    ref = [dw_call(True), dw_call(False), ln_call()]
    arena = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    _lib.launch("gt_defer_begin", _p(arena), arena.numel())
    got = [dw_call(True), dw_call(False), ln_call()]
    torch.cuda.synchronize()
    assert float(got[0][0][0, 0]) == 7.0 and float(got[2][1][0]) == 7.0    # nothing summed yet
    with pytest.raises(RuntimeError):
        _lib.launch("gt_defer_end")                                         # queued sums were never flushed
    _lib.launch("gt_defer_flush", st)
    _lib.launch("gt_defer_end")
    torch.cuda.synchronize()
    for r, g in zip(ref[:2], got[:2]):
        assert torch.equal(r[0], g[0]) and torch.equal(r[1], g[1])
    assert torch.equal(ref[2][0], got[2][0])
    assert rel(got[2][1], ref[2][1]) < 1e-5 and rel(got[2][2], ref[2][2]) < 1e-5
    # an arena too small for the partials: the producer reduces on the spot
    small = torch.empty(4096, dtype=torch.uint8, device=DEV)
    _lib.launch("gt_defer_begin", _p(small), small.numel())
    dw, db, _ = dw_call(True)
    _lib.launch("gt_defer_flush", st)
    _lib.launch("gt_defer_end")
    assert torch.equal(dw, ref[0][0]) and torch.equal(db, ref[0][1])


@pytest.mark.parametrize("tok_dtype,tol", [(torch.bfloat16, 1e-2), (torch.float32, 1e-5)])
@pytest.mark.parametrize("max_len,with_cls", [(1000, 1), (9, 1), (60, 0)])
def test_row_map_with_layernorm_equals_gather_then_layernorm(tok_dtype, tol, max_len, with_cls):
    """gt_linear_set_rows_layernorm + gt_seq_token_rows_layernorm (gnn2transformer + pad + CLS + norm_input as one GEMM,
    models/gnn_transformer.py:92-96, modules/transformer_encoder.py:50-57) against gt_linear_fwd -> gt_seq_gather_cls32 ->
    gt_layernorm_fwd: the un-normalised token rows bit for bit, the normalised rows / saved statistics to one rounding of the storage type."""
    import numpy as np
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    from graphtrans_amd.w3 import W3Images
    lib = _lib.lib()
    rng = np.random.default_rng(max_len + 7)
    B, K, N = 33, 600, 128
    sizes = rng.integers(1, 150, B)
    Nn = int(sizes.sum())
    assert Nn >= 1024 and lib.gt_linear_rows_layernorm_ok(N) == 1 and lib.gt_linear_rows_layernorm_ok(100) == 0
    gptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    kept = np.minimum(sizes, min(int(sizes.max()), max_len))
    kv = kept + with_cls
    tok_ptr = np.concatenate([[0], np.cumsum(kv)])
    desc = np.zeros((B, 4), np.int32)
    desc[:, 0], desc[:, 1], desc[:, 3] = tok_ptr[:-1], kv, kv
    rows = int(tok_ptr[-1])
    d_gptr, d_desc = torch.tensor(gptr, device=DEV), torch.tensor(desc, device=DEV)
    d_ng = torch.tensor(np.repeat(np.arange(B, dtype=np.int32), sizes), device=DEV)
    torch.manual_seed(2)
    x = torch.randn(Nn, K, device=DEV) * (0.5 + torch.rand(Nn, 1, device=DEV))
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    bias, cls = torch.randn(N, device=DEV), torch.randn(N, device=DEV)
    lw, lb = torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV) * 0.2
    tcode = GT_BF16 if tok_dtype == torch.bfloat16 else GT_F32
    imgs = W3Images([W])
    imgs.build()
    st = _stream()
    with imgs.bound():
        hn = torch.empty(Nn, N, dtype=tok_dtype, device=DEV)
        _lib.launch("gt_linear_fwd_ld2", GT_F32, tcode, GT_F32, _p(x), _p(W), _p(bias), _p(hn), Nn, N, K, K, N, 0, 0.0, 0, st)
        tok_ref = torch.zeros(rows, N, dtype=tok_dtype, device=DEV)
        _lib.launch("gt_seq_gather_cls32", tcode, _p(hn), _p(cls), _p(d_gptr), _p(d_desc), B, 1, int(kv.max()), with_cls, N, _p(tok_ref), st)
        xin_ref = torch.empty_like(tok_ref)
        mean_ref, rstd_ref = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
        _lib.launch("gt_layernorm_fwd", tcode, _p(tok_ref), None, _p(lw), _p(lb), 1e-5, 0.0, 0, rows, N, _p(xin_ref), _p(mean_ref), _p(rstd_ref), st)
        tok, xin = torch.zeros(rows, N, dtype=tok_dtype, device=DEV), torch.zeros(rows, N, dtype=tok_dtype, device=DEV)
        mean, rstd = torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV)
        rmap = torch.empty(Nn, dtype=torch.int32, device=DEV)
        _lib.launch("gt_seq_token_rows_layernorm", tcode, _p(cls) if with_cls else None, _p(d_gptr), _p(d_ng), _p(d_desc), B, 1, with_cls, Nn, N, _p(tok),
                    _p(rmap), _p(lw), _p(lb), 1e-5, _p(xin), _p(mean), _p(rstd), st)
        _lib.launch("gt_linear_set_rows_layernorm", _p(rmap), _p(lw), _p(lb), 1e-5, _p(xin), _p(mean), _p(rstd))
        _lib.launch("gt_linear_fwd_ld2", GT_F32, tcode, GT_F32, _p(x), _p(W), _p(bias), _p(tok), Nn, N, K, K, N, 0, 0.0, 0, st)
    assert torch.equal(tok, tok_ref)
    assert rel(mean, mean_ref) < 1e-5 and rel(rstd, rstd_ref) < 1e-5
    assert float((xin.float() - xin_ref.float()).abs().max()) <= tol * max(1.0, float(xin_ref.float().abs().max()))
    assert rel(xin.float(), xin_ref.float()) < tol * 0.1


@pytest.mark.parametrize("M,T,K,N", [(15945, 4, 68, 136), (15945, 4, 340, 204), (1500, 3, 36, 20), (4097, 2, 100, 324)])
def test_grouped_towers_on_bound_images(M, T, K, N):
    """gt_linear_fwd_grouped / gt_linear_bwd_grouped with every tower weight bound (PNAConv's pre / post tower stacks,
    modules/pna/pna_module.py:116-133, at the Code2-PNA shapes): k_lin3 with blockIdx.y = tower, column slices in and out, bias, the two
    dX addends; held to the float64 evaluation like the one-group kernel, and the weight gradient to the exact kernel's bits."""
    from graphtrans_amd import _lib, w3
    from graphtrans_amd.graph import _stream
    torch.manual_seed(M + K)
    x = torch.randn(M, T * K, device=DEV)
    W = (torch.randn(T, N, K, device=DEV) / K ** 0.5).contiguous()
    b = torch.randn(T, N, device=DEV)
    dy = torch.randn(M, T * N, device=DEV)
    a1, a2 = torch.randn(M, T * K, device=DEV), torch.randn(M, T * K, device=DEV)
    imgs = w3.W3Images([W[t] for t in range(T)])
    imgs.build()
    ws_bytes = _lib.lib().gt_linear_bwd_grouped_workspace_bytes(GT_F32, M, N, K, T)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=DEV)

    def run():
        y = torch.empty(M, T * N, device=DEV)
        dx = torch.empty(M, T * K, device=DEV)
        dw, db = torch.empty(T, N, K, device=DEV), torch.empty(T, N, device=DEV)
        _lib.launch("gt_linear_fwd_grouped", GT_F32, GT_F32, GT_F32, _p(x), _p(W), _p(b), _p(y), M, N, K, T * K, T * N, T, K, N, 0, 0.0, 0, _stream())
        _lib.launch("gt_linear_bwd_grouped", GT_F32, GT_F32, GT_F32, _p(x), _p(W), _p(dy), None, _p(a1), _p(a2), _p(dx), _p(dw), _p(db),
                    M, N, K, T * K, T * N, T, K, N, 0.0, _p(ws), ws_bytes, _stream())
        torch.cuda.synchronize()
        return y, dx, dw, db

    y0, dx0, dw0, db0 = run()            # exact-fp32 tiled kernels
    with imgs.bound():
        y1, dx1, dw1, db1 = run()        # grouped bf16x6
    xd, Wd = x.double().view(M, T, K), W.double()
    yr = (torch.einsum("mtk,tnk->mtn", xd, Wd) + b.double()).reshape(M, T * N)
    dxr = torch.einsum("mtn,tnk->mtk", dy.double().view(M, T, N), Wd).reshape(M, T * K) + a1.double() + a2.double()
    assert rel(y1, yr) <= 4e-7 and rel(dx1, dxr) <= 4e-7
    assert rel(y1, yr) <= 2.0 * rel(y0, yr) + 1e-7 and rel(dx1, dxr) <= 2.0 * rel(dx0, dxr) + 1e-7
    assert float((y1.double() - yr).abs().max()) <= 2e-5 * float(yr.abs().max())
    dwr = torch.einsum("mtn,mtk->tnk", dy.double().view(M, T, N), xd)
    assert rel(dw1, dwr) <= 1e-6 and rel(db1, dy.double().view(M, T, N).sum(0)) <= 1e-6
