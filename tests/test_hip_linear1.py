"""Weight-stationary bf16 GEMMs of the encoder layers (csrc/linear1.h) through the C ABI (gt_w1_images / gt_w1_bind +
gt_linear_fwd_ld2 / gt_linear_bwd_ld2 / gt_linear_bwd_gate_out) against a float64 evaluation of the SAME bf16 operands, with the tiled
bf16 kernels (unbound weight) as the yardstick, at the encoder shapes of the benchmarked configurations: d_model 128 / ffn 512
(Code2, Molpcba, PNA), ffn 256 (NCI1), d_model 256 / ffn 1024 (ER), ragged and tiny M."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GT_F32, GT_BF16 = 0, 1
BF = torch.bfloat16


def _p(t):
    return None if t is None else t.data_ptr()


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def fwd(x, W, b, bound, act=0, p=0.0, seed=0):
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty(M, N, dtype=BF, device=x.device)
    call = lambda: _lib.launch("gt_linear_fwd_ld2", GT_BF16, GT_BF16, GT_BF16, _p(x), _p(W), _p(b), _p(y), M, N, K, K, N, act, p, seed, _stream())
    if bound is not None:
        with bound.bound():
            call()
    else:
        call()
    return y


def bwd(x, W, dy, ymask, add1, add2, bound, p=0.0, gate_out=False, want_dw=False):
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    M, N = dy.shape
    K = W.shape[1]
    dx = torch.empty(M, K, dtype=BF, device=dy.device)
    dw = torch.zeros(N, K, device=dy.device) if want_dw else None
    db = torch.zeros(N, device=dy.device) if want_dw else None
    ws_bytes = _lib.lib().gt_linear_bwd_workspace_bytes(GT_BF16, M, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device)
    name = "gt_linear_bwd_gate_out" if gate_out else "gt_linear_bwd_ld2"
    call = lambda: _lib.launch(name, GT_BF16, GT_BF16, GT_BF16, _p(x), _p(W), _p(dy), _p(ymask), _p(add1), _p(add2), _p(dx), _p(dw), _p(db),
                               M, N, K, K, N, p, _p(ws), ws_bytes, _stream())
    if bound is not None:
        with bound.bound():
            call()
    else:
        call()
    return (dx, dw, db) if want_dw else dx


# (M, N, K): forward shapes; their dX runs the (K, N) image
SHAPES = [(31598, 384, 128), (31598, 128, 128), (31598, 512, 128), (31598, 128, 512),   # Code2 / Molpcba / PNA encoder
          (1153, 256, 128), (1153, 128, 256),                                         # NCI1 ffn 256
          (20011, 256, 256), (20011, 512, 256), (20011, 256, 512),   # d_model 256
          (63, 384, 128), (64, 128, 512), (1, 512, 128), (4099, 128, 128)]


@pytest.mark.parametrize("M,N,K", SHAPES, ids=[f"{m}x{n}x{k}" for m, n, k in SHAPES])
def test_forward_and_dx_match_float64_of_the_same_bf16_operands(M, N, K):
    from graphtrans_amd import _lib
    from graphtrans_amd.w3 import W1Images
    lib = _lib.lib()
    assert lib.gt_w1_image_bytes(N, K) > 0 and lib.gt_w1_image_bytes(K, N) > 0
    torch.manual_seed(M + N + K)
    x = (torch.randn(M, K, device=DEV) * (1.0 + 3.0 * torch.rand(M, 1, device=DEV))).to(BF)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    imgs = W1Images([W])
    imgs.build()
    y1, y0 = fwd(x, W, b, imgs), fwd(x, W, b, None)
    Wb = W.to(BF).double()
    y64 = x.double() @ Wb.t() + b.double()
    e1, e0 = rel(y1, y64), rel(y0, y64)
    print(f"\nfwd {M}x{N}x{K}: stationary {e1:.2e}  tiled {e0:.2e}")
    assert e1 <= 1.05 * e0 + 1e-5 and e1 < 4e-3      # one bf16 rounding of the output: ~2^-9 / sqrt(3)
    assert rel(y1, y0) < 4e-3
    # relu + dropout: the SAME mask as the tiled kernel (the hash is a function of (seed, row, column) only)
    r1, r0 = fwd(x, W, b, imgs, act=1, p=0.3, seed=1234567), fwd(x, W, b, None, act=1, p=0.3, seed=1234567)
    assert bool(((r1 == 0) == (r0 == 0)).all()) or float(((r1 == 0) != (r0 == 0)).float().mean()) < 1e-4   # (values within rounding of 0 may flip)
    ref = torch.relu(y64) / 0.7 * (r0 != 0)
    assert rel(r1, ref) <= 1.05 * rel(r0, ref) + 1e-5
    # dX = dY W + add1 + add2
    dy = torch.randn(M, N, device=DEV).to(BF)
    a1, a2 = torch.randn(M, K, device=DEV).to(BF), torch.randn(M, K, device=DEV).to(BF)
    d1, d0 = bwd(x, W, dy, None, a1, a2, imgs), bwd(x, W, dy, None, a1, a2, None)
    d64 = dy.double() @ Wb + a1.double() + a2.double()
    e1, e0 = rel(d1, d64), rel(d0, d64)
    print(f"dx  {M}x{N}x{K}: stationary {e1:.2e}  tiled {e0:.2e}")
    assert e1 <= 1.05 * e0 + 1e-5 and e1 < 4e-3
    d1n = bwd(x, W, dy, None, None, None, imgs)
    assert rel(d1n, dy.double() @ Wb) < 4e-3


@pytest.mark.parametrize("M,N,K", [(31598, 128, 512), (4099, 128, 256), (20011, 256, 1024)], ids=["code2-l2", "nci1-l2", "er-l2"])
def test_gate_on_the_dx_output_is_the_gated_gradient_both_gemms_of_linear1_read(M, N, K):
    """linear2's backward (x = f1 [M][K], W2 [N][K], dy = dF2 [M][N]) with the ReLU / dropout gate of f1 on its dX OUTPUT:
    dZ1 = (dF2 W2) * 1[f1 > 0] / keep; dW2 / db2 from the un-gated dy.  And the multiplier form (dropout_p < 0)."""
    from graphtrans_amd import _lib
    from graphtrans_amd.w3 import W1Images
    lib = _lib.lib()
    torch.manual_seed(N + K)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    f1 = (torch.relu(torch.randn(M, K, device=DEV)) * (torch.rand(M, K, device=DEV) > 0.3)).to(BF)
    dy = torch.randn(M, N, device=DEV).to(BF)
    imgs = W1Images([W])
    imgs.build()
    with imgs.bound():
        assert lib.gt_linear_bwd_gate_out_ok(GT_BF16, GT_BF16, GT_BF16, W.data_ptr(), M, N, K) == 1
    assert lib.gt_linear_bwd_gate_out_ok(GT_BF16, GT_BF16, GT_BF16, W.data_ptr(), M, N, K) == 0   # unbound
    dz, dw, db = bwd(f1, W, dy, f1, None, None, imgs, p=0.3, gate_out=True, want_dw=True)
    Wb = W.to(BF).double()
    ref = (dy.double() @ Wb) * (f1 > 0) / 0.7
    assert rel(dz, ref) < 4e-3
    assert bool((dz[f1 <= 0] == 0).all())
    dw64, db64 = dy.double().t() @ f1.double(), dy.double().sum(0)
    assert rel(dw, dw64) < 2e-3 and rel(db, db64) < 1e-4
    # the tiled path gates dY of the NEXT call instead: same tensor within one extra bf16 rounding
    mul = torch.rand(M, K, device=DEV).to(BF)
    dzm = bwd(f1, W, dy, mul, None, None, imgs, p=-1.0, gate_out=True)
    assert rel(dzm, (dy.double() @ Wb) * mul.double()) < 4e-3
    # unbound: the call reports that it is not covered instead of mis-gating
    from graphtrans_amd.graph import _stream
    dx = torch.empty(M, K, dtype=BF, device=DEV)
    ws_bytes = lib.gt_linear_bwd_workspace_bytes(GT_BF16, M, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    rc = lib.gt_linear_bwd_gate_out(GT_BF16, GT_BF16, GT_BF16, None, _p(W), _p(dy), _p(f1), None, None, _p(dx), None, None, M, N, K, K, N, 0.3,
                                    _p(ws), ws_bytes, _stream())
    assert rc != 0


def test_uncovered_shapes_keep_the_tiled_kernels():
    from graphtrans_amd import _lib
    from graphtrans_amd.w3 import W1Images
    lib = _lib.lib()
    assert lib.gt_w1_image_bytes(300, 128) == 0 and lib.gt_w1_image_bytes(128, 96) == 0 and lib.gt_w1_image_bytes(128, 2048) == 0
    torch.manual_seed(0)
    M, N, K = 5000, 128, 96     # forward not covered (K % 128), dX (image of W^T: rows 96) not covered either
    x = torch.randn(M, K, device=DEV).to(BF)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    imgs = W1Images([W])
    imgs.build()
    y1, y0 = fwd(x, W, None, imgs), fwd(x, W, None, None)
    assert torch.equal(y1, y0)


@pytest.mark.parametrize("M,N,K,p", [(31598, 128, 128, 0.3), (31598, 128, 512, 0.3), (4099, 128, 256, 0.0), (20011, 256, 256, 0.1), (61, 128, 128, 0.3)],
                         ids=["out_proj", "linear2", "nci1-l2", "d256", "tiny"])
def test_gemm_with_layernorm_epilogue_equals_the_two_kernels(M, N, K, p):
    """gt_linear_layernorm_fwd (a = x W^T + b saved; y = LayerNorm(resid + dropout(a)); mean / rstd saved) against gt_linear_fwd
    (tiled kernel) + gt_layernorm_fwd, and against float64 of the same bf16 operands with the SAME dropout mask."""
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    from graphtrans_amd.w3 import W1Images
    lib = _lib.lib()
    torch.manual_seed(M + K)
    x = torch.randn(M, K, device=DEV).to(BF)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    resid = (2.0 * torch.randn(M, N, device=DEV)).to(BF)
    g, be = 1.0 + 0.1 * torch.randn(N, device=DEV), 0.1 * torch.randn(N, device=DEV)
    imgs = W1Images([W])
    imgs.build()
    a1, y1 = torch.empty(M, N, dtype=BF, device=DEV), torch.empty(M, N, dtype=BF, device=DEV)
    mu1, rs1 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    seed = 0x1234567890ABCDEF
    assert lib.gt_linear_layernorm_fwd_ok(GT_BF16, GT_BF16, _p(W), M, N, K) == 0   # unbound
    with imgs.bound():
        assert lib.gt_linear_layernorm_fwd_ok(GT_BF16, GT_BF16, _p(W), M, N, K) == 1
        _lib.launch("gt_linear_layernorm_fwd", GT_BF16, GT_BF16, _p(x), _p(W), _p(b), _p(a1), M, N, K, _p(resid), _p(g), _p(be), 1e-5, p, seed,
                    _p(y1), _p(mu1), _p(rs1), _stream())
    a0 = fwd(x, W, b, None)
    y0 = torch.empty(M, N, dtype=BF, device=DEV)
    mu0, rs0 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    _lib.launch("gt_layernorm_fwd", GT_BF16, _p(a0), _p(resid), _p(g), _p(be), 1e-5, p, seed, M, N, _p(y0), _p(mu0), _p(rs0), _stream())
    assert rel(a1, a0) < 4e-3
    # the LayerNorm of the fused kernel on ITS saved a, by the stand-alone kernel: same rows up to the reduction order
    y2 = torch.empty(M, N, dtype=BF, device=DEV)
    mu2, rs2 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    _lib.launch("gt_layernorm_fwd", GT_BF16, _p(a1), _p(resid), _p(g), _p(be), 1e-5, p, seed, M, N, _p(y2), _p(mu2), _p(rs2), _stream())
    assert float((mu1 - mu2).abs().max()) < 1e-5 * max(1.0, float(mu2.abs().max())) and rel(rs1, rs2) < 1e-5
    assert float((y1.float() - y2.float()).abs().max()) <= 2 ** -7 * float(y2.float().abs().max()) and rel(y1, y2) < 1e-3   # <= one bf16 ulp apart
    assert rel(y1, y0) < 1e-2
    # float64 with the mask the kernels used (recovered from the stand-alone kernel at weight 1 / bias 0 is overkill: p = 0 case only)
    if p == 0.0:
        z = resid.double() + a1.double()
        ref = torch.nn.functional.layer_norm(z, (N,), g.double(), be.double(), 1e-5)
        assert rel(y1, ref) < 4e-3
        assert rel(mu1, z.mean(1)) < 1e-5 and rel(rs1, 1.0 / torch.sqrt(z.var(1, unbiased=False) + 1e-5)) < 1e-5


@pytest.mark.parametrize("M,N,K,p,resid,adds", [(31598, 512, 128, 0.3, True, 1), (31598, 384, 128, 0.3, True, 1), (31598, 128, 128, 0.0, True, 2),
                                                   (4099, 512, 128, 0.3, False, 0), (61, 384, 128, 0.3, True, 1)],
                         ids=["linear1", "in_proj", "out_proj-shape", "no-resid", "tiny"])
def test_dx_gemm_with_layernorm_backward_epilogue_equals_the_two_kernels(M, N, K, p, resid, adds):
    """gt_linear_bwd_dx_layernorm (g = dY W + addends never stored; d_sub, d_resid and the LayerNorm weight / bias gradients out of the
    GEMM's epilogue) against gt_linear_bwd_ld2 (dX) + gt_layernorm_bwd on the stored bf16 g (post-norm encoder layer backward,
    modules/transformer_encoder.py:28-32), same dropout mask; row sums in another order: <= a bf16 ulp on the rows, 1e-4 on the sums."""
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    from graphtrans_amd.w3 import W1Images
    lib = _lib.lib()
    torch.manual_seed(M + N)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    dy = torch.randn(M, N, device=DEV).to(BF)
    a1 = torch.randn(M, K, device=DEV).to(BF) if adds >= 1 else None
    a2 = torch.randn(M, K, device=DEV).to(BF) if adds >= 2 else None
    sub = torch.randn(M, K, device=DEV).to(BF)                       # the sub-layer output the forward normalised
    res = (2.0 * torch.randn(M, K, device=DEV)).to(BF) if resid else None
    gam = 1.0 + 0.1 * torch.randn(K, device=DEV)
    seed = 0x0FEDCBA987654321
    mu, rs = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    y = torch.empty(M, K, dtype=BF, device=DEV)
    _lib.launch("gt_layernorm_fwd", GT_BF16, _p(sub), _p(res), _p(gam), _p(torch.zeros(K, device=DEV)), 1e-5, p, seed, M, K, _p(y), _p(mu), _p(rs), _stream())
    imgs = W1Images([W])
    imgs.build()
    assert lib.gt_linear_bwd_dx_layernorm_ok(GT_BF16, GT_BF16, _p(W), M, N, K) == 0   # unbound
    ds1, dr1 = torch.empty(M, K, dtype=BF, device=DEV), torch.empty(M, K, dtype=BF, device=DEV)
    dg1, db1 = torch.empty(K, device=DEV), torch.empty(K, device=DEV)
    ws_bytes = int(lib.gt_linear_bwd_dx_layernorm_workspace_bytes(M, N, K))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    with imgs.bound():
        assert lib.gt_linear_bwd_dx_layernorm_ok(GT_BF16, GT_BF16, _p(W), M, N, K) == 1
        _lib.launch("gt_linear_bwd_dx_layernorm", GT_BF16, GT_BF16, _p(W), _p(dy), _p(a1), _p(a2), M, N, K, _p(sub), _p(res), _p(gam), _p(mu), _p(rs),
                    p, seed, _p(ds1), _p(dr1), _p(dg1), _p(db1), _p(ws), ws_bytes, _stream())
        g = bwd(None, W, dy, None, a1, a2, imgs)           # the same weight-stationary dX GEMM, g stored
    ds0, dr0 = torch.empty(M, K, dtype=BF, device=DEV), torch.empty(M, K, dtype=BF, device=DEV)
    dg0, db0 = torch.empty(K, device=DEV), torch.empty(K, device=DEV)
    lws_bytes = int(lib.gt_layernorm_bwd_workspace_bytes(M, K))
    lws = torch.empty(lws_bytes, dtype=torch.uint8, device=DEV)
    _lib.launch("gt_layernorm_bwd", GT_BF16, _p(sub), _p(res), _p(g), _p(gam), _p(mu), _p(rs), p, seed, M, K, _p(ds0), _p(dr0), _p(dg0), _p(db0),
                _p(lws), lws_bytes, _stream())
    torch.cuda.synchronize()
    for got, want in ((dr1, dr0), (ds1, ds0)):
        assert float((got.float() - want.float()).abs().max()) <= 2 ** -7 * float(want.float().abs().max()) and rel(got, want) < 1e-3
    assert rel(dg1, dg0) < 1e-4 and rel(db1, db0) < 1e-4
    if p > 0:
        assert bool(((ds1 == 0) == (ds0 == 0)).all())     # the same dropout mask
    # only d_resid (norm_input's use: no sub-layer gradient)
    dr2 = torch.empty(M, K, dtype=BF, device=DEV)
    with imgs.bound():
        _lib.launch("gt_linear_bwd_dx_layernorm", GT_BF16, GT_BF16, _p(W), _p(dy), _p(a1), _p(a2), M, N, K, _p(sub), _p(res), _p(gam), _p(mu), _p(rs),
                    p, seed, None, _p(dr2), _p(dg1), _p(db1), _p(ws), ws_bytes, _stream())
    torch.cuda.synchronize()
    assert torch.equal(dr2, dr1)


@pytest.mark.parametrize("M,N,K", [(31598, 384, 128), (31598, 128, 128), (31598, 512, 128), (31598, 128, 512), (20011, 1024, 256), (5003, 72, 200), (1024, 136, 8),
                                    (1025, 128, 128)], ids=lambda v: str(v))
def test_bf16_weight_gradient(M, N, K):
    """dW = dY^T X, db = colsum(dY) for bf16 rows (M-splits + fixed-order reduce) against float64 of the same bf16 operands, at the
    encoder shapes and at ragged ones; twice the same bits (no atomics)."""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV).to(BF)
    dy = (torch.randn(M, N, device=DEV) * (0.5 + torch.rand(M, 1, device=DEV))).to(BF)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    _, dw, db = bwd(x, W, dy, None, None, None, None, want_dw=True)
    dw64, db64 = dy.double().t() @ x.double(), dy.double().sum(0)
    assert rel(dw, dw64) < 2e-6 and rel(db, db64) < 2e-6      # exact bf16 products, fp32 accumulation
    _, dw2, db2 = bwd(x, W, dy, None, None, None, None, want_dw=True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("M,N,K,ldy,ldx", [(8197, 128, 128, 384, 256), (4096 + 77, 256, 128, 264, 136), (1024, 128, 256, 128, 512)], ids=lambda v: str(v))
def test_bf16_weight_gradient_strided_operands(M, N, K, ldy, ldx):
    """The LDS-DMA ring kernel (csrc/linear_dw16.h) on column slices of wider buffers (the q / k / v slices of qkv, one half of a
    concatenation): pitches ldy > N and ldx > K, a ragged last stage, dW only / dW + db; against float64 of the same bf16 operands."""
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    torch.manual_seed(M + ldy)
    xb = torch.randn(M, ldx, device=DEV).to(BF)
    dyb = torch.randn(M, ldy, device=DEV).to(BF)
    x, dy = xb[:, ldx - K:], dyb[:, ldy - N:]          # the LAST K / N columns: the base pointers are offset too (16-byte aligned)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    ws_bytes = _lib.lib().gt_linear_bwd_workspace_bytes(GT_BF16, M, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    dw64, db64 = dy.double().t() @ x.double(), dy.double().sum(0)
    for want_db in (True, False):
        dw = torch.zeros(N, K, device=DEV)
        db = torch.zeros(N, device=DEV) if want_db else None
        _lib.launch("gt_linear_bwd_ld2", GT_BF16, GT_BF16, GT_BF16, x.data_ptr(), _p(W), dy.data_ptr(), None, None, None, None, _p(dw), _p(db),
                    M, N, K, ldx, ldy, 0.0, _p(ws), ws_bytes, _stream())
        assert rel(dw, dw64) < 2e-6
        assert float((dw.double() - dw64).abs().max()) < 1e-3 * float(dw64.abs().max())
        if want_db:
            assert rel(db, db64) < 2e-6
