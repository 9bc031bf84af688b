"""GPU parity of gt_linear_fwd / gt_linear_bwd against plain fp32 PyTorch (F.linear [+relu]) on the
shapes of the hot path.  fp32 compute (exact-fp32 MFMA) 1e-4; bf16 compute 3e-2 with operands
rounded to bf16 on both sides."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(31598, 300, 300), (31855, 384, 128), (31855, 512, 128), (31855, 128, 512), (1000, 128, 600), (256, 600, 300), (256, 300, 600), (512, 1024, 2048),
          (77, 12, 20), (129, 132, 68), (5, 4, 4)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("mode", ["fp32", "bf16c_fp32s", "bf16"])
@pytest.mark.parametrize("act", [None, "relu", "gelu"])
def test_linear_fwd_bwd(M, N, K, mode, act):
    from graphtrans_amd import ops

    if mode == "bf16" and (K % 8 or N % 8):
        pytest.skip("bf16 storage needs 16-byte rows (K, N multiples of 8): the kernels reject other shapes (ops.linear has no "
                    "fallback GEMM), and no bf16-storage GEMM of the path has one")
    torch.manual_seed(0)
    x = torch.randn(M, K)
    w = torch.randn(N, K) / K ** 0.5
    b = torch.randn(N) * 0.1
    g = torch.randn(M, N)
    if mode != "fp32":  # both sides see bf16-rounded operands
        x, w_ref, g = x.bfloat16().float(), w.bfloat16().float(), g.bfloat16().float()
    else:
        w_ref = w
    xr, wr, br = x.double().requires_grad_(True), w_ref.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.linear(xr, wr, br)
    yr = F.relu(yr) if act == "relu" else (F.gelu(yr) if act == "gelu" else yr)   # F.gelu: the erf form, as in torch 1.7
    (yr * g.double()).sum().backward()
    ops.set_matmul_dtype(torch.float32 if mode == "fp32" else torch.bfloat16)
    try:
        sdt = torch.bfloat16 if mode == "bf16" else torch.float32
        xd = x.to(DEV).to(sdt).requires_grad_(True)
        wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        yd = ops.linear(xd, wd, bd, act=act)
        (yd.float() * g.to(DEV)).sum().backward()
    finally:
        ops.set_matmul_dtype(torch.float32)
    tol = 1e-4 if mode == "fp32" else 3e-2
    y_ref, gx_ref, gw_ref, gb_ref = yr.detach(), xr.grad, wr.grad, br.grad
    gx, gw, gb = xd.grad.float().cpu(), wd.grad.cpu(), bd.grad.cpu()
    if act == "relu":  # relu gate ties / near-zero pre-activations flip under different rounding: exclude them
        z = F.linear(x.double(), w_ref.double(), b.double())
        tie = z.abs() < (1e-4 if mode == "fp32" else 5e-2)
        if mode == "fp32":
            assert tie.float().mean() < 0.01
        # recompute the reference gradients with the ties zeroed on both sides
        gmask = g.double().masked_fill(tie, 0.0)
        xr2, wr2, br2 = x.double().requires_grad_(True), w_ref.double().requires_grad_(True), b.double().requires_grad_(True)
        (F.relu(F.linear(xr2, wr2, br2)) * gmask).sum().backward()
        xd2 = x.to(DEV).to(sdt).requires_grad_(True)
        wd2, bd2 = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        ops.set_matmul_dtype(torch.float32 if mode == "fp32" else torch.bfloat16)
        try:
            (ops.linear(xd2, wd2, bd2, act=act).float() * gmask.float().to(DEV)).sum().backward()
        finally:
            ops.set_matmul_dtype(torch.float32)
        gx_ref, gw_ref, gb_ref = xr2.grad, wr2.grad, br2.grad
        gx, gw, gb = xd2.grad.float().cpu(), wd2.grad.cpu(), bd2.grad.cpu()
    assert_close(yd.float().cpu(), y_ref, atol=tol, rtol=tol, what="y")
    assert_close(gx, gx_ref, atol=tol, rtol=tol, what="dx")
    assert_close(gw, gw_ref, atol=tol, rtol=tol, what="dW")
    assert_close(gb, gb_ref, atol=tol, rtol=tol, what="db")


def test_linear_gelu_fused_dropout():
    """gelu + dropout in the epilogue; the backward multiplies by the saved gelu'(z) * mask / (1 - p)."""
    from graphtrans_amd import ops

    torch.manual_seed(2)
    M, N, K, p = 3000, 512, 128, 0.3
    for sdt in (torch.float32, torch.bfloat16):
        x = torch.randn(M, K, device=DEV).to(sdt).requires_grad_(True)
        w = (torch.randn(N, K, device=DEV) / K ** 0.5).requires_grad_(True)
        b = (torch.randn(N, device=DEV) * 0.1).requires_grad_(True)
        y = ops.linear(x, w, b, act="gelu", dropout_p=p, seed=11)
        y0 = ops.linear(x, w, b, act="gelu")
        keep = (y != 0) | (y0 == 0)
        assert abs(keep.float().mean().item() - (1 - p)) < 0.01
        assert torch.allclose(y[keep].float(), y0[keep].float() / (1 - p), rtol=2e-2 if sdt == torch.bfloat16 else 1e-5, atol=1e-6)
        g = torch.randn(M, N, device=DEV)
        (y.float() * g).sum().backward()
        xr = x.detach().double().cpu().requires_grad_(True)
        wr = (w.detach().bfloat16() if sdt == torch.bfloat16 else w.detach()).double().cpu().requires_grad_(True)
        br = b.detach().double().cpu().requires_grad_(True)
        yr = F.gelu(F.linear(xr, wr, br)) * keep.cpu().double() / (1 - p)
        (yr * g.cpu().double()).sum().backward()
        tol = 3e-2 if sdt == torch.bfloat16 else 1e-4
        assert_close(x.grad.float().cpu(), xr.grad, atol=tol, rtol=tol, what="dx")
        assert_close(w.grad.cpu(), wr.grad, atol=tol, rtol=tol, what="dW")
        assert_close(b.grad.cpu(), br.grad, atol=tol, rtol=tol, what="db")
        assert torch.equal(ops.linear(x, w, b, act="gelu", dropout_p=p, seed=11), y)


def test_linear_fused_dropout():
    """relu + dropout fused in the epilogue; backward recovers the mask from y > 0."""
    from graphtrans_amd import ops

    torch.manual_seed(1)
    M, N, K, p = 4096, 512, 128, 0.3
    x = torch.randn(M, K, device=DEV, requires_grad=True)
    w = (torch.randn(N, K, device=DEV) / K ** 0.5).requires_grad_(True)
    b = torch.zeros(N, device=DEV, requires_grad=True)
    y = ops.linear(x, w, b, act="relu", dropout_p=p, seed=7)
    y0 = ops.linear(x, w, b, act="relu")
    pos = y0 > 0
    kept = (y > 0)[pos].float().mean().item()
    assert abs(kept - (1 - p)) < 0.01
    m = (y > 0)
    assert torch.allclose(y[m], y0[m] / (1 - p), rtol=1e-5)
    g = torch.randn(M, N, device=DEV)
    (y * g).sum().backward()
    xr, wr = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True)
    yr = F.relu(F.linear(xr, wr)) * m.cpu().double() / (1 - p)
    (yr * g.cpu().double()).sum().backward()
    assert_close(x.grad.cpu(), xr.grad, what="dx")
    assert_close(w.grad.cpu(), wr.grad, what="dW")
    y2 = ops.linear(x, w, b, act="relu", dropout_p=p, seed=7)
    assert torch.equal(y2, y)


@pytest.mark.parametrize("M,N,K", [(256, 25010, 128), (64, 5002, 128), (33, 10, 8), (300, 2050, 64), (300, 4102, 128), (17, 4097, 128), (513, 8200, 128)])
@pytest.mark.parametrize("mode", ["fp32", "bf16c_fp32s"])
def test_linear_padded_rows_and_split_dx(M, N, K, mode):
    """gt_linear_*_ld: an N that is not a multiple of 4 lands in row-padded storage (the 5 x 5002-way
    stacked heads, models/gnn_transformer.py:124-126); the long-contraction dX is split over N."""
    from graphtrans_amd import ops

    torch.manual_seed(1)
    x = torch.randn(M, K)
    w = torch.randn(N, K) / K ** 0.5
    b = torch.randn(N) * 0.1
    g = torch.randn(M, N)
    if mode != "fp32":
        x, w_ref, g = x.bfloat16().float(), w.bfloat16().float(), g.bfloat16().float()
    else:
        w_ref = w
    xr, wr, br = x.double().requires_grad_(True), w_ref.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.linear(xr, wr, br)
    (yr * g.double()).sum().backward()
    ops.set_matmul_dtype(torch.float32 if mode == "fp32" else torch.bfloat16)
    try:
        xd = x.to(DEV).requires_grad_(True)
        wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        ld = (N + 3) // 4 * 4
        yd = ops.linear(xd, wd, bd, ldy=ld)
        assert yd.shape == (M, N) and yd.stride() == (ld, 1)
        (yd * g.to(DEV)).sum().backward()
    finally:
        ops.set_matmul_dtype(torch.float32)
    tol = 1e-4 if mode == "fp32" else 3e-2
    assert_close(yd.detach().cpu(), yr.detach(), atol=tol, rtol=tol, what="y")
    assert_close(xd.grad.cpu(), xr.grad, atol=tol, rtol=tol, what="dx")
    assert_close(wd.grad.cpu(), wr.grad, atol=tol, rtol=tol, what="dw")
    assert_close(bd.grad.cpu(), br.grad, atol=tol, rtol=tol, what="db")


@pytest.mark.parametrize("M,N,K", [(200, 64, 37), (33, 6, 5), (1, 128, 1)])
def test_linear_module_odd_K(M, N, K):
    """K not a multiple of the 16-byte chunk (TU node encoder, 37 features): x and W are zero-padded along K by
    gt_repitch; outputs and all three gradients equal the plain fp32 GEMM (no torch fallback)."""
    from graphtrans_amd import ops
    torch.manual_seed(0)
    lin = torch.nn.Linear(K, N).cuda()
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    y = ops.linear_module(lin, x)
    g = torch.randn(M, N, device="cuda")
    y.backward(g)
    xr = x.detach().double().requires_grad_()
    w, b = lin.weight.detach().double().requires_grad_(), lin.bias.detach().double().requires_grad_()
    yr = torch.nn.functional.linear(xr, w, b)
    yr.backward(g.double())
    assert y.shape == (M, N)
    for got, want in ((y, yr), (x.grad, xr.grad), (lin.weight.grad, w.grad), (lin.bias.grad, b.grad)):
        assert got.shape == want.shape
        assert (got.double() - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
    with pytest.raises(RuntimeError):
        ops.linear_module(lin.cpu(), x.detach().cpu())


@pytest.mark.parametrize("M,T,K,N,bias", [(1000, 4, 68, 68, True), (257, 4, 340, 204, True), (64, 1, 16, 8, False), (5000, 3, 32, 100, True)])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_tower_linear_grouped(M, T, K, N, bias, mode):
    """gt_linear_fwd/bwd_grouped (grid.y = tower): PNAConv's per-tower pre_nns / post_nns (modules/pna/pna_module.py:33-41),
    y[:, t] = x[:, t] W[t]^T + b[t], against float64 einsum; all gradients."""
    from graphtrans_amd import ops
    torch.manual_seed(M + T)
    ops.set_matmul_dtype(torch.bfloat16 if mode == "bf16" else torch.float32)
    try:
        x = torch.randn(M, T, K, device="cuda", requires_grad=True)
        w = torch.randn(T, N, K, device="cuda", requires_grad=True)
        b = torch.randn(T, N, device="cuda", requires_grad=True) if bias else None
        y = ops.tower_linear(x, w, b)
        g = torch.randn(M, T, N, device="cuda")
        y.backward(g)
        xr, wr = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
        br = b.detach().double().requires_grad_() if bias else None
        if mode == "bf16":   # the kernel rounds both operands to bf16 (fp32 accumulate)
            xq, wq = xr.float().bfloat16().double(), wr.float().bfloat16().double()
            xq, wq = xr + (xq - xr).detach(), wr + (wq - wr).detach()
        else:
            xq, wq = xr, wr
        yr = torch.einsum("mtk,tnk->mtn", xq, wq) + (br if bias else 0)
        yr.backward(g.double())
        tol = 2e-2 if mode == "bf16" else 1e-4
        pairs = [(y, yr), (x.grad, xr.grad), (w.grad, wr.grad)] + ([(b.grad, br.grad)] if bias else [])
        for got, want in pairs:
            assert got.shape == want.shape
            assert (got.double() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    finally:
        ops.set_matmul_dtype(torch.float32)


@pytest.mark.parametrize("sdt", [torch.float32, torch.bfloat16])
def test_weight_gradient_only_entry_points(sdt):
    """gt_linear_bwd_dw_forked / gt_linear_bwd_mul_dw_forked (dW and db alone; outside an overlap section they run on the
    caller's stream) give the bits of the full backward's dW / db -- ReLU-gated and multiplier (GELU) forms."""
    import ctypes as C
    from graphtrans_amd import _lib

    lib = _lib.lib()
    torch.manual_seed(4)
    M, N, K = 2500, 384, 128
    code = _lib.GT_BF16 if sdt == torch.bfloat16 else _lib.GT_F32
    comp = _lib.GT_BF16 if sdt == torch.bfloat16 else _lib.GT_F32
    x = torch.randn(M, K, device=DEV).to(sdt)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    dy = torch.randn(M, N, device=DEV).to(sdt)
    y = torch.relu(torch.randn(M, N, device=DEV)).to(sdt)          # forward output: gate = y > 0
    gm = (torch.rand(M, N, device=DEV) * 1.5).to(sdt)               # saved multiplier
    ws_bytes = int(lib.gt_linear_bwd_workspace_bytes(comp, M, N, K))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def P(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def run(fn, *args):
        _lib.check(getattr(lib, fn)(*args), fn)

    dx = torch.empty_like(x)
    for mul in (False, True):
        dw0, db0 = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        dw1, db1 = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        if mul:
            run("gt_linear_bwd_mul", code, code, comp, P(x), P(w), P(dy), P(gm), None, None, P(dx), P(dw0), P(db0), M, N, K, K, N,
                P(ws), ws_bytes, st)
            run("gt_linear_bwd_mul_dw_forked", code, code, comp, P(x), P(w), P(dy), P(gm), P(dw1), P(db1), M, N, K, K, N, P(ws),
                ws_bytes, st)
            z = dy.float() * gm.float()
        else:
            run("gt_linear_bwd", code, code, comp, P(x), P(w), P(dy), P(y), None, None, P(dx), P(dw0), P(db0), M, N, K, 0.0,
                P(ws), ws_bytes, st)
            run("gt_linear_bwd_dw_forked", code, code, comp, P(x), P(w), P(dy), P(y), P(dw1), P(db1), M, N, K, K, N, 0.0, P(ws),
                ws_bytes, st)
            z = dy.float() * (y > 0).float()
        torch.cuda.synchronize()
        assert torch.equal(dw0, dw1) and torch.equal(db0, db1), ("mul" if mul else "relu")
        tol = 3e-2 if sdt == torch.bfloat16 else 1e-4
        assert_close(dw1.cpu(), (z.double().t() @ x.double()).float().cpu(), atol=tol, rtol=tol, what="dW")
        assert_close(db1.cpu(), z.double().sum(0).float().cpu(), atol=tol, rtol=tol, what="db")
