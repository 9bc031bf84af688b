"""GPU parity of the BatchNorm / LayerNorm kernels (gt_batchnorm_*, gt_layernorm_*) against plain
fp32 PyTorch references of the same ops (CPU, float64 accumulate where it matters).
fp32 tolerance 1e-4 (scale-relative); bf16 storage 3e-2."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows,D", [(257, 300), (5000, 600), (4, 16), (32407, 300), (300, 2048), (1024, 600), (1025, 36), (2, 4),
                                     (3000, 300), (8192, 300), (8193, 300), (6611, 600)])   # a few thousand rows: Molpcba, Code2 at 32 graphs per rank
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_matches_torch(rows, D, relu, training):
    from graphtrans_amd.modules.norm import BatchNorm1d

    torch.manual_seed(0)
    x = torch.randn(rows, D) * 2.0 + torch.linspace(-50, 50, D)  # |mean| >> std on some columns
    w = torch.randn(rows, D)
    ref = torch.nn.BatchNorm1d(D).double()
    with torch.no_grad():
        ref.weight.copy_(torch.randn(D) * 0.5 + 1)
        ref.bias.copy_(torch.randn(D) * 0.3)
        ref.running_mean.copy_(torch.randn(D))
        ref.running_var.copy_(torch.rand(D) + 0.5)
    m = BatchNorm1d(D)
    m.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
    m = m.to(DEV)
    ref.train(training)
    m.train(training)
    if relu:  # the ReLU gate of an output within fp32 rounding of 0 is a coin flip: give those elements no upstream grad
        with torch.no_grad():
            tie = ref(x.double()).abs() < 1e-4
            ref.running_mean.copy_(m.running_mean.double().cpu())  # undo the extra stat update of the probe call
            ref.running_var.copy_(m.running_var.double().cpu())
            ref.num_batches_tracked.copy_(m.num_batches_tracked.cpu())
        w = w.masked_fill(tie, 0.0)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr = F.relu(yr) if relu else yr
    (yr * w.double()).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    yd = m(xd, relu=relu)
    (yd * w.to(DEV)).sum().backward()
    assert_close(yd.cpu(), yr.detach(), what="y")
    assert_close(xd.grad.cpu(), xr.grad, what="dx")
    assert_close(m.weight.grad.cpu(), ref.weight.grad, what="dweight")
    assert_close(m.bias.grad.cpu(), ref.bias.grad, what="dbias")
    assert_close(m.running_mean.cpu(), ref.running_mean, what="running_mean")
    assert_close(m.running_var.cpu(), ref.running_var, what="running_var")
    assert int(m.num_batches_tracked) == int(ref.num_batches_tracked)


@pytest.mark.parametrize("rows,D", [(1000, 128), (33, 16), (777, 256), (50, 1024), (31855, 128)])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("with_resid", [True, False])
def test_layernorm_matches_torch(rows, D, dtype, tol, with_resid):
    from graphtrans_amd import ops

    torch.manual_seed(1)
    x = (torch.randn(rows, D) * 1.5 + 0.3).to(dtype).float()
    r = torch.randn(rows, D).to(dtype).float() if with_resid else None
    wt = torch.randn(rows, D)
    g, b = torch.randn(D) * 0.5 + 1, torch.randn(D) * 0.2
    xr = x.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if with_resid else None
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.layer_norm(xr + rr if with_resid else xr, (D,), gr, br, 1e-5)
    (yr * wt.double()).sum().backward()
    xd = x.to(DEV).to(dtype).requires_grad_(True)
    rd = r.to(DEV).to(dtype).requires_grad_(True) if with_resid else None
    gd, bd = g.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    yd = ops.layer_norm(xd, gd, bd, 1e-5, resid=rd)
    (yd.float() * wt.to(DEV)).sum().backward()
    assert_close(yd.float().cpu(), yr.detach(), atol=tol, rtol=tol, what="y")
    assert_close(xd.grad.float().cpu(), xr.grad, atol=tol, rtol=tol, what="dx")
    if with_resid:
        assert_close(rd.grad.float().cpu(), rr.grad, atol=tol, rtol=tol, what="dresid")
    assert_close(gd.grad.cpu(), gr.grad, atol=tol, rtol=tol, what="dweight")
    assert_close(bd.grad.cpu(), br.grad, atol=tol, rtol=tol, what="dbias")


def test_layernorm_dropout_replay():
    """Dropout inside the fused LN is a pure function of (seed,row,col): recover the mask from a
    probe, then check forward/backward against torch with that explicit mask."""
    from graphtrans_amd import ops

    torch.manual_seed(2)
    rows, D, p, seed = 512, 128, 0.3, 99
    g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    # probe: resid = 0, x = 1 + small ramp -> z = mask/(1-p) * x ; mask = (z != 0) via a non-normalising trick:
    x = torch.randn(rows, D)
    r = torch.randn(rows, D)
    # recover the mask from the backward: dx = dz * mask/(1-p), dresid = dz
    xd = x.to(DEV).requires_grad_(True)
    rd = r.to(DEV).requires_grad_(True)
    y = ops.layer_norm(xd, g, b, 1e-5, resid=rd, dropout_p=p, seed=seed)
    wt = torch.randn(rows, D, device=DEV)
    (y * wt).sum().backward()
    ratio = (xd.grad / rd.grad).cpu()
    mask = ratio.abs() > 1e-6
    assert abs(mask.float().mean().item() - (1 - p)) < 0.02
    assert torch.allclose(ratio[mask], torch.full_like(ratio[mask], 1 / (1 - p)), rtol=1e-4)
    xr = x.double().requires_grad_(True)
    rr = r.double().requires_grad_(True)
    yr = F.layer_norm(xr * mask.double() / (1 - p) + rr, (D,), None, None, 1e-5)
    (yr * wt.cpu().double()).sum().backward()
    assert_close(y.detach().cpu(), yr.detach(), what="y (dropout)")
    assert_close(xd.grad.cpu(), xr.grad, what="dx (dropout)")
    assert_close(rd.grad.cpu(), rr.grad, what="dresid (dropout)")
    y2 = ops.layer_norm(xd.detach(), g, b, 1e-5, resid=rd.detach(), dropout_p=p, seed=seed)
    y3 = ops.layer_norm(xd.detach(), g, b, 1e-5, resid=rd.detach(), dropout_p=p, seed=seed + 1)
    assert torch.equal(y2, y.detach()) and not torch.equal(y3, y.detach())


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("rows,D", [(1000, 300), (257, 64), (5000, 300), (2000, 64)])
def test_batchnorm_fused_dropout(rows, D, relu):
    """F.dropout behind the layer BatchNorm (gnn_module.py:88-90,209-212) fused into gt_batchnorm_*: the
    kept set is replayed by the backward; values are bn(x)[relu] / (1 - p) or 0."""
    from graphtrans_amd import ops
    torch.manual_seed(3)
    p = 0.3
    x = torch.randn(rows, D, device=DEV).requires_grad_(True)
    w = (torch.rand(D, device=DEV) + 0.5).requires_grad_(True)
    b = torch.randn(D, device=DEV).mul_(0.2).requires_grad_(True)
    rm, rv, nbt = torch.zeros(D, device=DEV), torch.ones(D, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    y = ops.batch_norm(x, w, b, rm, rv, nbt, 0.1, 1e-5, True, relu, dropout_p=p, seed=1234567)
    y0 = ops.batch_norm(x.detach(), w.detach(), b.detach(), rm.clone(), rv.clone(), nbt.clone(), 0.1, 1e-5, True, relu)
    live = y0 != 0
    kept = (y != 0) & live
    rate = 1.0 - kept.sum().item() / live.sum().item()
    assert abs(rate - p) < 0.01, rate
    assert torch.allclose(y[kept], y0[kept] / (1 - p), rtol=1e-6, atol=1e-7)
    assert (y[~kept] == 0).all()
    # same seed -> same mask; another seed -> another mask
    y_again = ops.batch_norm(x.detach(), w.detach(), b.detach(), rm.clone(), rv.clone(), nbt.clone(), 0.1, 1e-5, True, relu,
                             dropout_p=p, seed=1234567)
    assert torch.equal(y_again, y.detach())
    y_other = ops.batch_norm(x.detach(), w.detach(), b.detach(), rm.clone(), rv.clone(), nbt.clone(), 0.1, 1e-5, True, relu,
                             dropout_p=p, seed=7654321)
    assert not torch.equal(y_other, y.detach())
    # backward: autograd of the same function with the recovered mask, in float64
    g = torch.randn(rows, D, device=DEV)
    y.backward(g)
    xr, wr, br = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    mu, var = xr.mean(0), xr.var(0, unbiased=False)
    yr = (xr - mu) / torch.sqrt(var + 1e-5) * wr + br
    if relu:
        yr = torch.relu(yr)
    (yr * kept.double() / (1 - p) * g.double()).sum().backward()
    assert_close(x.grad.double().cpu(), xr.grad.cpu(), atol=1e-4, rtol=1e-4, what="dx")
    assert_close(w.grad.double().cpu(), wr.grad.cpu(), atol=1e-4, rtol=1e-4, what="dw")
    assert_close(b.grad.double().cpu(), br.grad.cpu(), atol=1e-4, rtol=1e-4, what="db")
    # eval mode: dropout is off
    ye = ops.batch_norm(x.detach(), w.detach(), b.detach(), rm, rv, None, 0.1, 1e-5, False, relu, dropout_p=p, seed=1)
    ye0 = ops.batch_norm(x.detach(), w.detach(), b.detach(), rm, rv, None, 0.1, 1e-5, False, relu)
    assert torch.equal(ye, ye0)


def _bn_keep_mask(rows, D, p, seed):
    """the kept set of the BatchNorm dropout (csrc/norm.hip: bn_hash of (seed, row, column) >= p * 2^32), restated with int64 tensors"""
    M = 0xFFFFFFFF
    r = torch.arange(rows, dtype=torch.int64).reshape(-1, 1)
    c = torch.arange(D, dtype=torch.int64).reshape(1, -1)
    s0, s1 = seed & M, (seed >> 32) & M
    x = (((r * 0x9E3779B1) & M) + s0 & M) ^ (((c * 0x85EBCA77) & M) + s1 & M)
    x = x ^ (x >> 16); x = (x * 0x7feb352d) & M
    x = x ^ (x >> 15); x = (x * 0x846ca68b) & M
    x = x ^ (x >> 16)
    thr = min(max(int(p * 4294967296.0), 1), 4294967295)
    return x >= thr



# (> 1024 rows: Code2's shape with ~10 graphs per block, 8000 tiny graphs, wide and narrow rows)
@pytest.mark.parametrize("rows,D,B", [(1500, 300, 40), (6611, 600, 256), (8192, 128, 7), (31598, 300, 256), (20000, 300, 8000), (46000, 64, 900)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_mid_rows_full_epilogue(rows, D, B, relu):
    """gt_batchnorm_fwd_bcast at a few thousand rows (Molpcba, Code2 at 32 graphs per rank): relu + dropout + residual + the next layer's
    virtual-node rows (gnn_module.py:199,204-212) against float64 torch with the kernel's own dropout mask, and the backward
    of the same call."""
    import ctypes as C
    from graphtrans_amd import _lib
    torch.manual_seed(11)
    p, seed, eps, mom = 0.25, 99, 1e-5, 0.1
    x = (torch.randn(rows, D) * 1.5 + torch.linspace(-20, 20, D)).to(DEV)
    w = (torch.rand(D) + 0.5).to(DEV)
    b = (torch.randn(D) * 0.2).to(DEV)
    resid = torch.randn(rows, D, device=DEV)
    vn = torch.randn(B, D, device=DEV)
    idx = torch.sort(torch.randint(0, B, (rows,), dtype=torch.int32)).values.to(DEV)
    rm, rv = torch.zeros(D, device=DEV), torch.ones(D, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    y = torch.empty_like(x)
    mean, rstd = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    wsb = _lib.lib().gt_batchnorm_workspace_bytes(rows, D)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.launch("gt_batchnorm_fwd_bcast", 0, ptr(x), ptr(w), ptr(b), ptr(rm), ptr(rv), ptr(nbt), mom, eps, 1, int(relu), ptr(resid),
                ptr(vn), ptr(idx), None, rows, D, ptr(y), ptr(mean), ptr(rstd), p, seed, ptr(ws), wsb, st)
    # the same call without dropout / addends gives the mask
    y0 = torch.empty_like(x)
    rm2, rv2, mean2, rstd2 = rm.clone(), rv.clone(), torch.empty_like(mean), torch.empty_like(rstd)   # (named: temporaries would share one block)
    _lib.launch("gt_batchnorm_fwd_bcast", 0, ptr(x), ptr(w), ptr(b), ptr(rm2), ptr(rv2), None, mom, eps, 1, int(relu), None,
                None, None, None, rows, D, ptr(y0), ptr(mean2), ptr(rstd2), 0.0, 0, ptr(ws), wsb, st)
    add = resid + vn[idx.long()]
    kept = _bn_keep_mask(rows, D, p, seed).to(DEV)
    clear = y0.abs() > 1e-3   # (a kept value that small may vanish in the sum with the addends)
    assert torch.equal(((y - add) != 0)[clear], kept[clear])
    xd = x.double().cpu()
    mu, var = xd.mean(0), xd.var(0, unbiased=False)
    yr = (xd - mu) / torch.sqrt(var + eps) * w.double().cpu() + b.double().cpu()
    if relu:   # the kernel's own gate (an output within fp32 rounding of 0 is a coin flip)
        yr = yr * (y0 > 0).double().cpu()
    yr = yr * kept.double().cpu() / (1 - p) + add.double().cpu()
    rate = 1.0 - kept.double().mean().item()
    assert abs(rate - p) < 0.02, rate
    assert_close(y.cpu().double(), yr, atol=1e-4, rtol=1e-4, what="y")
    assert_close(mean.cpu().double(), mu, atol=1e-4, rtol=1e-5, what="mean")
    assert_close(rm.cpu().double(), mom * mu, atol=1e-4, rtol=1e-5, what="running_mean")
    assert_close(rv.cpu().double(), (1 - mom) + mom * xd.var(0, unbiased=True), atol=1e-4, rtol=1e-4, what="running_var")
    assert int(nbt) == 1
    # backward of the same call
    g = torch.randn(rows, D, device=DEV)
    dx, dw, db = torch.empty_like(x), torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    _lib.launch("gt_batchnorm_bwd", 0, ptr(x), ptr(g), ptr(w), ptr(b), ptr(mean), ptr(rstd), 1, int(relu), rows, D, ptr(dx), ptr(dw), ptr(db),
                p, seed, ptr(ws), wsb, st)
    xr = xd.clone().requires_grad_(True)
    wr, br = w.double().cpu().requires_grad_(True), b.double().cpu().requires_grad_(True)
    m2, v2 = xr.mean(0), xr.var(0, unbiased=False)
    z = (xr - m2) / torch.sqrt(v2 + eps) * wr + br
    if relu:
        z = z * (y0 > 0).double().cpu()
    (z * kept.double().cpu() / (1 - p) * g.double().cpu()).sum().backward()
    assert_close(dx.cpu().double(), xr.grad, atol=1e-4, rtol=1e-4, what="dx")
    assert_close(dw.cpu().double(), wr.grad, atol=1e-4, rtol=1e-4, what="dw")
    assert_close(db.cpu().double(), br.grad, atol=1e-4, rtol=1e-4, what="db")


