"""Linear + BatchNorm over few rows as one launch per direction (csrc/linear_bn_slab.h: the virtual-node MLP of
modules/gnn_module.py:161-170 on one row per graph) against float64 torch, through the C ABI.

forward : z = x W^T + b, y = [dropout(relu(BN(z)))] (+ resid), saved statistics, running statistics -- the dropout mask is recovered
          from the separate BatchNorm kernel run with the same seed (bn_hash(seed, row, column): the fused kernel must draw the same).
backward: d a = dz_up W_up through the backward of the BatchNorm(+ReLU) below -> dz, d gamma, d beta (autograd float64).
Shapes: the virtual-node MLP's (256 x 600 x 300, 256 x 300 x 600), ragged row / column counts, the 512-row and 2-row ends."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(256, 600, 300), (256, 300, 600), (37, 20, 12), (512, 128, 96), (2, 4, 4), (130, 300, 600), (33, 44, 100)]


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("relu", [0, 1])
@pytest.mark.parametrize("extras", [False, True])   # dropout + residual
def test_slab_forward_against_float64(M, N, K, relu, extras):
    from graphtrans_amd import _lib
    L = _lib.lib()
    assert L.gt_linear_bn_slab_ok(0, M, N, K, 1) == 1
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K)
    w = torch.randn(N, K) / K ** 0.5
    b = torch.randn(N) * 0.3 + torch.linspace(-20, 20, N)   # |mean| >> std on some columns
    bw, bb = torch.rand(N) + 0.5, torch.randn(N) * 0.2
    rm, rv = torch.randn(N), torch.rand(N) + 0.5
    resid = torch.randn(M, N) if extras else None
    p, seed, eps, mom = (0.25, 1234567, 1e-5, 0.1) if extras else (0.0, 0, 1e-5, 0.1)
    xd, wd, bd, bwd, bbd = (t.to(DEV) for t in (x, w, b, bw, bb))
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    z, y = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    mean, rstd = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    rd = resid.to(DEV) if resid is not None else None
    _lib.launch("gt_linear_bn_slab_fwd", 0, _p(xd), _p(wd), _p(bd), _p(z), _p(bwd), _p(bbd), _p(rmd), _p(rvd), _p(nbt), mom, eps, relu, _p(rd),
                M, N, K, _p(y), _p(mean), _p(rstd), p, seed, _st())
    torch.cuda.synchronize()
    z64 = x.double() @ w.double().t() + b.double()
    mu = z64.mean(0)
    var = z64.var(0, unbiased=False)
    assert_close(z.cpu(), z64, what="z")
    assert_close(mean.cpu(), mu, what="mean")
    assert_close(rstd.cpu(), 1.0 / torch.sqrt(var + eps), atol=1e-4, rtol=1e-4, what="rstd")
    assert_close(rmd.cpu(), (1 - mom) * rm.double() + mom * mu, what="running_mean")
    assert_close(rvd.cpu(), (1 - mom) * rv.double() + mom * z64.var(0, unbiased=True), what="running_var")
    assert int(nbt) == 1
    yr = (z64 - mu) / torch.sqrt(var + eps) * bw.double() + bb.double()
    if relu:
        yr = F.relu(yr)
    if extras:   # the mask of the stand-alone BatchNorm kernel with the same seed (all-ones input through an identity BatchNorm)
        ones = torch.ones(M, N, device=DEV) + torch.arange(M, device=DEV, dtype=torch.float32)[:, None]   # rows differ: var > 0
        probe = torch.empty(M, N, device=DEV)
        ws_bytes = L.gt_batchnorm_workspace_bytes(M, N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        m2, r2 = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        one, zero = torch.ones(N, device=DEV), torch.full((N,), 1e3, device=DEV)   # y = xhat + 1000 > 0 wherever kept
        _lib.launch("gt_batchnorm_fwd", 0, _p(ones), _p(one), _p(zero), None, None, None, mom, eps, 1, 0, None, M, N, _p(probe), _p(m2), _p(r2),
                    p, seed, _p(ws), ws_bytes, _st())
        torch.cuda.synchronize()
        keep = (probe != 0).cpu()
        if M * N >= 400:
            assert 0.6 < float(keep.float().mean()) < 0.9
        yr = yr * keep.double() / (1 - p) + resid.double()
    assert_close(y.cpu(), yr, what="y")


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("relu", [0, 1])
def test_slab_backward_against_float64(M, N, K, relu):
    """upper Linear: a [M][K] -> N outputs (W_up [N][K]); below it: a = relu?(BN(z)), z [M][K]"""
    from graphtrans_amd import _lib
    L = _lib.lib()
    assert L.gt_linear_bn_slab_ok(0, M, N, K, 1) == 1
    torch.manual_seed(M * 3 + N + K)
    z = torch.randn(M, K) * 1.5 + torch.linspace(-5, 5, K)
    w_up = torch.randn(N, K) / K ** 0.5
    bw, bb = torch.rand(K) + 0.5, torch.randn(K) * 0.2
    dz_up = torch.randn(M, N)
    eps = 1e-5
    z64 = z.double().requires_grad_(True)
    bw64, bb64 = bw.double().requires_grad_(True), bb.double().requires_grad_(True)
    mu, var = z64.mean(0), z64.var(0, unbiased=False)
    pre = (z64 - mu) / torch.sqrt(var + eps) * bw64 + bb64
    if relu:   # gates within fp32 rounding of zero are coin flips: give those elements no upstream gradient
        tie = pre.detach().abs() < 1e-4
    a = F.relu(pre) if relu else pre
    up = a @ w_up.double().t()
    g_up = dz_up.double()
    if relu:
        # zero the upstream gradient reaching tied elements: d a = dz_up W_up, so mask d a directly through a surrogate loss
        da = (g_up @ w_up.double()).masked_fill(tie, 0.0)
        (a * da).sum().backward()
    else:
        (up * g_up).sum().backward()
    mean = mu.detach().float().to(DEV)
    rstd = (1.0 / torch.sqrt(var.detach() + eps)).float().to(DEV)
    dz = torch.empty(M, K, device=DEV)
    dg, db = torch.empty(K, device=DEV), torch.empty(K, device=DEV)
    dzu_d, wu_d, z_d, bw_d, bb_d = (t.to(DEV) for t in (dz_up, w_up, z, bw, bb))   # (kept alive: the launch takes raw pointers)
    _lib.launch("gt_linear_bn_slab_bwd", 0, _p(dzu_d), _p(wu_d), _p(z_d), _p(mean), _p(rstd), _p(bw_d), _p(bb_d),
                relu, M, N, K, _p(dz), _p(dg), _p(db), _st())
    torch.cuda.synchronize()
    if relu and bool(tie.any()):   # the kernel cannot know about the surrogate mask: compare where no tie sits in the column's statistics
        cols = ~tie.any(0)
        if not bool(cols.any()):
            pytest.skip("every column holds a tied gate")
        assert_close(dz.cpu()[:, cols], z64.grad[:, cols], atol=2e-4, rtol=2e-4, what="dz")
        assert_close(dg.cpu()[cols], bw64.grad[cols], atol=2e-4, rtol=2e-4, what="dgamma")
        assert_close(db.cpu()[cols], bb64.grad[cols], atol=2e-4, rtol=2e-4, what="dbeta")
    else:
        assert_close(dz.cpu(), z64.grad, atol=2e-4, rtol=2e-4, what="dz")
        assert_close(dg.cpu(), bw64.grad, atol=2e-4, rtol=2e-4, what="dgamma")
        assert_close(db.cpu(), bb64.grad, atol=2e-4, rtol=2e-4, what="dbeta")


def test_slab_is_bitwise_reproducible_and_refuses_what_it_does_not_cover():
    from graphtrans_amd import _lib
    L = _lib.lib()
    assert L.gt_linear_bn_slab_ok(0, 513, 300, 300, 1) == 0     # more rows than a block's tiles
    assert L.gt_linear_bn_slab_ok(0, 256, 300, 300, 0) == 0     # eval mode keeps the separate kernels
    assert L.gt_linear_bn_slab_ok(0, 256, 302, 300, 1) == 0     # 16-byte rows
    assert L.gt_linear_bn_slab_ok(0, 1, 300, 300, 1) == 0       # BatchNorm needs two rows
    M, N, K = 256, 600, 300
    torch.manual_seed(0)
    x, w = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / K ** 0.5
    b, bw, bb = torch.randn(N, device=DEV), torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV)
    outs = []
    for _ in range(3):
        z, y = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        mean, rstd = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        _lib.launch("gt_linear_bn_slab_fwd", 0, _p(x), _p(w), _p(b), _p(z), _p(bw), _p(bb), None, None, None, 0.1, 1e-5, 1, None, M, N, K, _p(y),
                    _p(mean), _p(rstd), 0.0, 0, _st())
        torch.cuda.synchronize()
        outs.append((z.clone(), y.clone(), mean.clone(), rstd.clone()))
    for o in outs[1:]:
        for a_, c_ in zip(outs[0], o):
            assert torch.equal(a_, c_)
