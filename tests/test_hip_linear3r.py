"""k_lin3r (csrc/linear3r.h: the bf16x6 GEMM with the activation rows loaded straight into MFMA fragments) against a float64 evaluation,
torch's fp32 GEMM as the yardstick (same 3 x bar as the other fp32-accurate kernels), through the C ABI.  Reference op: the nn.Linear of
GCNConv / GINConv on the node rows and its input gradient (modules/conv.py:44,51).  The kernel takes fp32 rows with M >= 12 288 and
8 < n-tiles per column block <= 20 when no gate / dropout / row map is asked for; the shapes below cover one and two column blocks, both
tile widths (NTW = 10 / 8), ragged M (not a multiple of 128), K tails (K % 32 = 12, 4, 16), narrow pitches and the epilogue options."""
import pytest
import torch

from test_hip_linear3x import dx_of, fwd, rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(31598, 300, 300), (16001, 272, 272), (20000, 256, 256), (13000, 600, 300), (12800, 304, 36), (12289, 132, 68), (12416, 320, 96)]


@pytest.mark.parametrize("M,N,K", SHAPES, ids=[f"{m}x{n}x{k}" for m, n, k in SHAPES])
def test_forward_bias_relu(M, N, K):
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV) * (1.0 + 3.0 * torch.rand(M, 1, device=DEV))
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    imgs = W3Images([W])
    imgs.build()
    y64 = torch.nn.functional.linear(x.double(), W.double(), b.double())
    t32 = torch.nn.functional.linear(x, W, b)
    et = rel(t32, y64)
    y3 = fwd(x, W, b, imgs)
    e3 = rel(y3, y64)
    print(f"\nfwd {M}x{N}x{K}: bf16x6 rows-in-registers {e3:.2e}  torch fp32 {et:.2e}")
    assert e3 <= max(3 * et, 1e-6), (e3, et)
    assert float((y3 - y64).abs().max()) <= 1e-4 * max(1.0, float(y64.abs().max()))
    # no bias, ReLU
    y3 = fwd(x, W, None, imgs, act=1)
    r64 = torch.relu(torch.nn.functional.linear(x.double(), W.double()))
    assert float((y3 - r64).abs().max()) <= 1e-4 * max(1.0, float(r64.abs().max()))
    # run to run bitwise identical
    assert torch.equal(fwd(x, W, None, imgs, act=1), y3)


@pytest.mark.parametrize("M,N,K", SHAPES, ids=[f"{m}x{n}x{k}" for m, n, k in SHAPES])
def test_dx_with_addends(M, N, K):
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(M + N + K + 1)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    dy = torch.randn(M, N, device=DEV) * (1.0 + 3.0 * torch.rand(M, 1, device=DEV))
    a1, a2 = torch.randn(M, K, device=DEV), torch.randn(M, K, device=DEV)
    imgs = W3Images([W])
    imgs.build()
    x = torch.empty(M, K, device=DEV)
    d64 = dy.double() @ W.double()
    et = rel(dy @ W, d64)
    d3 = dx_of(x, W, dy, None, None, None, imgs)
    e3 = rel(d3, d64)
    print(f"\ndX  {M}x{N}x{K}: bf16x6 rows-in-registers {e3:.2e}  torch fp32 {et:.2e}")
    assert e3 <= max(3 * et, 1e-6), (e3, et)
    d3 = dx_of(x, W, dy, None, a1, a2, imgs)
    ref = d64 + a1.double() + a2.double()
    assert rel(d3, ref) <= max(3 * rel(dy @ W + a1 + a2, ref), 1e-6)
    d3 = dx_of(x, W, dy, None, a1, None, imgs)
    assert float((d3 - (d64 + a1.double())).abs().max()) <= 1e-4 * max(1.0, float(d64.abs().max()))


def test_rows_past_m_and_columns_past_n_are_not_written():
    """the last block's rows beyond M and the last tile's columns beyond N: the output buffer around the result keeps its canary"""
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    from graphtrans_amd.w3 import W3Images
    torch.manual_seed(5)
    M, N, K = 12300, 300, 300
    x, W, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / K ** 0.5, torch.randn(N, device=DEV)
    imgs = W3Images([W])
    imgs.build()
    buf = torch.full((M + 200, N), 7.0, device=DEV)
    with imgs.bound():
        _lib.launch("gt_linear_fwd_ld2", 0, 0, 0, x.data_ptr(), W.data_ptr(), b.data_ptr(), buf.data_ptr(), M, N, K, K, N, 0, 0.0, 0, _stream())
    torch.cuda.synchronize()
    assert bool((buf[M:] == 7.0).all())
    y64 = torch.nn.functional.linear(x.double(), W.double(), b.double())
    assert float((buf[:M] - y64).abs().max()) <= 1e-4 * max(1.0, float(y64.abs().max()))
    # a wider output pitch: the columns between N and the pitch stay untouched
    wide = torch.full((M, N + 20), 7.0, device=DEV)
    with imgs.bound():
        _lib.launch("gt_linear_fwd_ld2", 0, 0, 0, x.data_ptr(), W.data_ptr(), b.data_ptr(), wide.data_ptr(), M, N, K, K, N + 20, 0, 0.0, 0, _stream())
    torch.cuda.synchronize()
    assert bool((wide[:, N:] == 7.0).all())
    assert float((wide[:, :N] - y64).abs().max()) <= 1e-4 * max(1.0, float(y64.abs().max()))


DW_SHAPES = [(31598, 300, 300), (31598, 384, 128), (31598, 128, 512), (6700, 600, 300), (1500, 20, 132), (12289, 132, 68), (33, 300, 300)]


@pytest.mark.parametrize("M,N,K", DW_SHAPES, ids=[f"{m}x{n}x{k}" for m, n, k in DW_SHAPES])
def test_pipelined_weight_gradient(M, N, K):
    """k_lin3r_dw (the stages of k_lin3_dw pipelined: two LDS buffers, the split of the next stage between the MFMAs of this one, db from
    running column sums) without and with a ReLU gate + keep scale, against float64; bitwise reproducible"""
    from graphtrans_amd.w3 import W3Images
    from test_hip_linear3x import bwd_all
    torch.manual_seed(M + 7 * N + K)
    x = torch.randn(M, K, device=DEV) * (0.5 + torch.rand(M, 1, device=DEV))
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    dy = torch.randn(M, N, device=DEV)
    imgs = W3Images([W])
    imgs.build()
    w64, b64 = dy.double().t() @ x.double(), dy.double().sum(0)
    wt, bt = dy.t() @ x, dy.sum(0)
    _, w3, b3 = bwd_all(x, W, dy, None, imgs)
    e3, et = rel(w3, w64), rel(wt, w64)
    print(f"\ndW {M}x{N}x{K}: pipelined bf16x6 {e3:.2e}  torch fp32 {et:.2e};  db {rel(b3, b64):.2e} / {rel(bt, b64):.2e}")
    assert e3 <= max(3 * et, 1e-6), (e3, et)
    assert rel(b3, b64) <= max(3 * rel(bt, b64), 1e-6)
    _, w3b, b3b = bwd_all(x, W, dy, None, imgs)
    assert torch.equal(w3, w3b) and torch.equal(b3, b3b)
    yf = torch.relu(torch.randn(M, N, device=DEV))
    dz = (dy.double() * (yf > 0)) / 0.8
    _, w3, b3 = bwd_all(x, W, dy, yf, imgs, p=0.2)
    dzf = (dy * (yf > 0)) / 0.8
    assert rel(w3, dz.t() @ x.double()) <= max(3 * rel(dzf.t() @ x, dz.t() @ x.double()), 1e-6)
    assert rel(b3, dz.sum(0)) <= max(3 * rel(dzf.sum(0), dz.sum(0)), 1e-6)


def test_dx_broadcast_addend():
    """gt_linear_bwd_bcast: dx[m] += rows[idx[m]] in the dX epilogue (the virtual-node update's gradient per graph,
    modules/gnn_module.py:219 backward) against the materialised addend; a call that cannot take the request fails loudly"""
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    from graphtrans_amd.w3 import W3Images
    L = _lib.lib()
    torch.manual_seed(21)
    M, N, K, B = 31598, 300, 300, 256
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    dy = torch.randn(M, N, device=DEV)
    a1 = torch.randn(M, K, device=DEV)
    rows = torch.randn(B, K, device=DEV)
    idx = torch.sort(torch.randint(0, B, (M,), dtype=torch.int32)).values.to(DEV)
    imgs = W3Images([W])
    imgs.build()
    x = torch.empty(M, K, device=DEV)
    ws_bytes = L.gt_linear_bwd_workspace_bytes(0, M, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    p = lambda t: None if t is None else t.data_ptr()

    def dx_call(add1, bc, m=M):
        dx = torch.empty(m, K, device=DEV)
        if bc:
            _lib.launch("gt_linear_bwd_bcast", p(rows), p(idx))
        _lib.launch("gt_linear_bwd_ld2", 0, 0, 0, None, p(W), p(dy), None, p(add1), None, p(dx), None, None, m, N, K, K, N, 0.0, p(ws), ws_bytes, _stream())
        return dx

    with imgs.bound():
        assert L.gt_linear_bwd_bcast_ok(0, 0, 0, p(W), M, N, K) == 1
        assert L.gt_linear_bwd_bcast_ok(0, 0, 0, p(W), 4096, N, K) == 0   # too few rows for the register-row kernel
        got = dx_call(a1, True)
        want = dx_call(a1 + rows[idx.long()], False)
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
        got2 = dx_call(None, True)
        ref = dy.double() @ W.double() + rows[idx.long()].double()
        assert float((got2 - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        assert torch.equal(dx_call(None, False), dx_call(None, False))   # the request does not outlive its call
        with pytest.raises(RuntimeError):
            dx_call(None, True, m=4096)
    assert L.gt_linear_bwd_bcast_ok(0, 0, 0, p(W), M, N, K) == 0   # nothing bound
