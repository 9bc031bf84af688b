"""LDS-ring bf16 GEMMs of the encoder layers on many token rows (csrc/linear2.h, k_lin2) through the C ABI (gt_w1_images / gt_w1_bind +
gt_linear_fwd_ld2 / gt_linear_bwd_ld2 / gt_linear_bwd_gate_out, option "lin_ring") against a float64 evaluation of the SAME bf16
operands, with the weight-stationary / tiled kernels ("lin_ring" = 1) as the yardstick: the Erdos-Renyi stress shapes (d_model 256,
ffn 1024: contraction 768 / 1024 does not fit k_lin1's registers), ragged row counts, 128- / 256- / 768- / 1024-column outputs."""
import pytest
import torch

from test_hip_linear1 import BF, DEV, GT_BF16, bwd, fwd, rel

pytestmark = pytest.mark.gpu


class ring:
    """`with ring(v):` runs the block with the library option "lin_ring" = v"""
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        from graphtrans_amd import _lib
        self.prev = _lib.option_set("lin_ring", self.v)

    def __exit__(self, *exc):
        from graphtrans_amd import _lib
        _lib.option_set("lin_ring", self.prev)


# (M, N, K): forward = (N rows, K contraction) image; dX = (K rows, N contraction) image
SHAPES = [(20011, 256, 1024), (20011, 256, 768), (20011, 1024, 256), (20011, 768, 256), (20011, 256, 256),   # ER encoder, either direction
          (2049, 128, 1024), (4096, 128, 512), (70001, 256, 1024), (66000, 1024, 256), (2048, 512, 128)]


@pytest.mark.parametrize("M,N,K", SHAPES, ids=[f"{m}x{n}x{k}" for m, n, k in SHAPES])
def test_ring_forward_and_dx_match_float64_of_the_same_bf16_operands(M, N, K):
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
    Wb = W.to(BF).double()
    y64 = x.double() @ Wb.t() + b.double()
    with ring(2):
        y2 = fwd(x, W, b, imgs)
        r2 = fwd(x, W, b, imgs, act=1, p=0.3, seed=1234567)
    with ring(1):
        y1 = fwd(x, W, b, imgs)
        r1 = fwd(x, W, b, imgs, act=1, p=0.3, seed=1234567)
    e2, e1 = rel(y2, y64), rel(y1, y64)
    print(f"\nfwd {M}x{N}x{K}: ring {e2:.2e}  yardstick {e1:.2e}")
    assert e2 <= 1.05 * e1 + 1e-5 and e2 < 4e-3      # one bf16 rounding of the output
    assert rel(y2, y1) < 4e-3
    # relu + dropout: the SAME mask (the hash is a function of (seed, row, column) only)
    assert float(((r2 == 0) != (r1 == 0)).float().mean()) < 1e-4   # (values within rounding of 0 may flip)
    ref = torch.relu(y64) / 0.7 * (r1 != 0)
    assert rel(r2, ref) <= 1.05 * rel(r1, ref) + 1e-5
    # dX = dY W + add1 + add2, dX = dY W
    dy = torch.randn(M, N, device=DEV).to(BF)
    a1, a2 = torch.randn(M, K, device=DEV).to(BF), torch.randn(M, K, device=DEV).to(BF)
    d64 = dy.double() @ Wb
    with ring(2):
        d2, d2n, d2a = bwd(x, W, dy, None, a1, a2, imgs), bwd(x, W, dy, None, None, None, imgs), bwd(x, W, dy, None, a1, None, imgs)
    with ring(1):
        d1 = bwd(x, W, dy, None, a1, a2, imgs)
    e2, e1 = rel(d2, d64 + a1.double() + a2.double()), rel(d1, d64 + a1.double() + a2.double())
    print(f"dx  {M}x{N}x{K}: ring {e2:.2e}  yardstick {e1:.2e}")
    assert e2 <= 1.05 * e1 + 1e-5 and e2 < 4e-3
    assert rel(d2n, d64) < 4e-3 and rel(d2a, d64 + a1.double()) < 4e-3


def test_default_policy_takes_the_ring_kernel_only_where_it_is_meant_to():
    """option 0: shapes k_lin1 does not cover (contraction 1024) take the ring kernel at any covered row count; covered ones keep k_lin1
    below 65 536 rows.  Observable through the launch profiler's kernel names."""
    from graphtrans_amd import _lib
    from graphtrans_amd.w3 import W1Images
    lib = _lib.lib()

    def names(M, N, K):
        x = torch.randn(M, K, device=DEV).to(BF)
        W = torch.randn(N, K, device=DEV) / K ** 0.5
        imgs = W1Images([W])
        imgs.build()
        lib.gt_profile_enable(32)
        fwd(x, W, None, imgs)
        torch.cuda.synchronize()
        import ctypes as C
        out = []
        for i in range(lib.gt_profile_count()):
            buf, ms, dims = C.create_string_buffer(64), C.c_float(), (C.c_int64 * 6)()
            lib.gt_profile_get(i, buf, 64, C.byref(ms), dims)
            out.append(buf.value.decode())
        lib.gt_profile_enable(0)
        return out

    assert _lib.option_get("lin_ring") == 0
    assert any(n.startswith("k_lin2") for n in names(4096, 256, 1024))
    assert any(n.startswith("k_lin1") for n in names(4096, 512, 128))
    assert any(n.startswith("k_lin2") for n in names(65536, 512, 128))
    assert any(n.startswith("k_lin1") for n in names(65536, 256, 256))    # out_proj at d_model 256: the weight-stationary kernel is as fast
    assert any(n.startswith("k_linear_fwd") for n in names(1024, 256, 1024))   # below 2 048 rows: the tiled kernel


@pytest.mark.parametrize("M", [20011, 70001], ids=["20k", "70k"])
def test_ring_gate_on_the_dx_output(M):
    """linear2's backward at the ER shape with the ReLU / dropout gate of f1 on its dX OUTPUT (gt_linear_bwd_gate_out): the ring kernel's
    gate / addend epilogue against float64, the multiplier form, and k_lin1 as the yardstick."""
    from graphtrans_amd.w3 import W1Images
    N, K = 256, 1024
    torch.manual_seed(N + K + M)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    f1 = (torch.relu(torch.randn(M, K, device=DEV)) * (torch.rand(M, K, device=DEV) > 0.3)).to(BF)
    dy = torch.randn(M, N, device=DEV).to(BF)
    add = torch.randn(M, K, device=DEV).to(BF)
    imgs = W1Images([W])
    imgs.build()
    Wb = W.to(BF).double()
    ref = (dy.double() @ Wb) * (f1 > 0) / 0.7
    with ring(2):
        dz = bwd(f1, W, dy, f1, None, None, imgs, p=0.3, gate_out=True)
        dza = bwd(f1, W, dy, f1, add, None, imgs, p=0.3, gate_out=True)
        mul = torch.rand(M, K, device=DEV).to(BF)
        dzm = bwd(f1, W, dy, mul, None, None, imgs, p=-1.0, gate_out=True)
    with ring(1):
        dz1 = bwd(f1, W, dy, f1, None, None, imgs, p=0.3, gate_out=True)
    assert rel(dz, ref) < 4e-3 and rel(dz, ref) <= 1.05 * rel(dz1, ref) + 1e-5
    assert bool((dz[f1 <= 0] == 0).all())
    assert rel(dza, ref + add.double()) < 4e-3
    assert rel(dzm, (dy.double() @ Wb) * mul.double()) < 4e-3


def test_ring_kernel_on_strided_operands():
    """Row pitches larger than the row length on both sides (column slices of wider buffers, as gt_linear_fwd_ld2 / _bwd_ld2 allow): the
    ring kernel's DMA source addresses and its epilogue's stores follow ldx / ldy; the columns outside the slice stay untouched."""
    from graphtrans_amd import _lib
    from graphtrans_amd.graph import _stream
    from graphtrans_amd.w3 import W1Images
    M, N, K, ldx, ldy = 4099, 256, 768, 1024, 384
    torch.manual_seed(5)
    xw = torch.randn(M, ldx, device=DEV).to(BF)
    x = xw[:, 128:128 + K]                       # 16-byte aligned column slice
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    imgs = W1Images([W])
    imgs.build()
    yw = torch.full((M, ldy), 7.0, dtype=BF, device=DEV)
    y = yw[:, 64:64 + N]
    with ring(2), imgs.bound():
        _lib.launch("gt_linear_fwd_ld2", GT_BF16, GT_BF16, GT_BF16, x.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, ldx, ldy, 0, 0.0, 0, _stream())
    torch.cuda.synchronize()
    ref = x.double() @ W.to(BF).double().t() + b.double()
    assert rel(y, ref) < 4e-3
    assert bool((yw[:, :64] == 7.0).all()) and bool((yw[:, 64 + N:] == 7.0).all())
    # dX = dY W on the same pitches (dY a slice of yw's shape, dX into a slice of a wide buffer), one addend
    dyw = torch.randn(M, ldy, device=DEV).to(BF)
    dy = dyw[:, 64:64 + N]
    dxw = torch.full((M, ldx), -3.0, dtype=BF, device=DEV)
    dx = dxw[:, 128:128 + K]
    addw = torch.randn(M, ldx, device=DEV).to(BF)
    add = addw[:, 128:128 + K]
    ws_bytes = _lib.lib().gt_linear_bwd_workspace_bytes(GT_BF16, M, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    with ring(2), imgs.bound():
        _lib.launch("gt_linear_bwd_ld2", GT_BF16, GT_BF16, GT_BF16, None, W.data_ptr(), dy.data_ptr(), None, add.data_ptr(), None, dx.data_ptr(), None, None,
                    M, N, K, ldx, ldy, 0.0, ws.data_ptr(), ws_bytes, _stream())
    torch.cuda.synchronize()
    assert rel(dx, dy.double() @ W.to(BF).double() + add.double()) < 4e-3
    assert bool((dxw[:, :128] == -3.0).all()) and bool((dxw[:, 128 + K:] == -3.0).all())
