"""gt_xent_fwd / _bwd (the Code2 loss over the stacked heads, dataset/code.py:39-45) against
torch.nn.CrossEntropyLoss per head in float64."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(stacked, y):
    L = stacked.shape[1]
    loss = 0
    for l in range(L):
        loss = loss + F.cross_entropy(stacked[:, l], y[:, l])
    return loss / L


@pytest.mark.parametrize("B,L,C,pad", [(256, 5, 5002, 2), (7, 3, 10, 2), (64, 1, 128, 0), (5, 5, 33, 3)])
@pytest.mark.parametrize("ignore", [False, True])
def test_xent_matches_per_head_cross_entropy(B, L, C, pad, ignore):
    from graphtrans_amd import ops
    torch.manual_seed(B + C)
    ld = L * C + pad
    buf = torch.randn(B, ld) * 3
    buf[:, L * C:] = float("nan")  # pad columns must never be read
    y = torch.randint(0, C, (B, L + 1))
    if ignore:
        y[::3, 0] = -100
        y[1, L - 1] = -100
    logits = buf[:, :L * C].reshape(B, L, C).double().requires_grad_(True)
    ref = _ref(logits, y)
    ref.backward()

    bd = buf.to(DEV)
    stacked = bd[:, :L * C].view(B, L, C).requires_grad_(True)
    loss = ops.softmax_xent(stacked, y.to(DEV))
    (loss * 2.5).backward()
    assert_close(loss.detach().cpu().double(), ref.detach(), atol=1e-5, rtol=1e-5, what="loss")
    assert_close(stacked.grad.cpu().double(), logits.grad * 2.5, atol=1e-6, rtol=1e-4, what="dlogits")
    # deterministic
    loss2 = ops.softmax_xent(stacked.detach(), y.to(DEV))
    assert torch.equal(loss2, loss.detach())


def test_code2_loss_uses_fused_path_and_matches_reference_formula():
    from graphtrans_amd import losses
    from graphtrans_amd.models.base_model import stacked_heads
    torch.manual_seed(0)
    heads = torch.nn.ModuleList([torch.nn.Linear(128, 5002) for _ in range(5)]).to(DEV)
    h = torch.randn(256, 128, device=DEV, requires_grad=True)
    y = torch.randint(0, 5002, (256, 5), device=DEV)
    preds = stacked_heads(h, heads, 5002)
    assert len(preds) == 5 and preds[0].shape == (256, 5002)
    loss = losses.code2_loss(preds, y)
    loss.backward()
    hr = h.detach().double().cpu().requires_grad_(True)
    ref = 0
    for i, m in enumerate(heads):
        ref = ref + F.cross_entropy(F.linear(hr, m.weight.detach().double().cpu(), m.bias.detach().double().cpu()), y[:, i].cpu())
    ref = ref / 5
    ref.backward()
    assert_close(loss.detach().cpu().double(), ref.detach(), atol=1e-4, rtol=1e-4, what="loss")
    assert_close(h.grad.cpu().double(), hr.grad, atol=1e-4, rtol=1e-4, what="dh")
    assert all(m.weight.grad is not None and m.bias.grad is not None for m in heads)


@pytest.mark.parametrize("B,T,ld", [(256, 128, 128), (7, 5, 8), (1, 3, 3), (300, 1000, 1000)])
def test_masked_bce_matches_oracle(B, T, ld):
    """dataset/mol.py:24-31 (oracle/reference_math.py:mol_loss): mean BCE-with-logits over the non-NaN labels."""
    from graphtrans_amd import losses, ops
    from oracle import reference_math as rm
    g = torch.Generator().manual_seed(B * 1000 + T)
    buf = torch.randn(B, ld, generator=g) * 3
    y = (torch.rand(B, T, generator=g) > 0.5).float()
    y[torch.rand(B, T, generator=g) < 0.6] = float("nan")
    if B > 2:
        y[1] = float("nan")        # a graph without any label
    y[0, 0] = 1.0                  # ... but never an empty selection (that case: test_masked_bce_no_labels_is_nan)
    pred_ref = buf[:, :T].clone().requires_grad_()
    want = rm.mol_loss(pred_ref, y)
    want.backward()
    dbuf = buf.cuda()
    pred = dbuf[:, :T].requires_grad_()       # rows padded to ld like the head GEMM's output
    got = losses.mol_loss(pred, y.cuda())
    got.backward()
    assert abs(got.item() - want.item()) <= 1e-5 * max(1.0, abs(want.item()))
    assert (pred.grad.cpu() - pred_ref.grad).abs().max().item() <= 1e-6
    # an explicit denominator (the data-parallel global count / world) rescales loss and gradient alike
    n = float((y == y).sum())
    p2 = dbuf[:, :T].detach().clone().requires_grad_()
    l2 = ops.masked_bce(p2, y.cuda(), torch.tensor([2.0 * n], device="cuda"))
    l2.backward()
    assert abs(l2.item() * 2 - want.item()) <= 1e-5 * max(1.0, abs(want.item()))
    assert (p2.grad.cpu() * 2 - pred_ref.grad).abs().max().item() <= 1e-6


def test_masked_bce_no_labels_is_nan():
    from graphtrans_amd import losses
    pred = torch.randn(4, 8, device="cuda", requires_grad=True)
    y = torch.full((4, 8), float("nan"), device="cuda")
    assert torch.isnan(losses.mol_loss(pred, y))


@pytest.mark.parametrize("B,C,ld", [(32, 2, 4), (5, 7, 7), (1, 2, 2)])
def test_tud_loss_is_the_one_head_cross_entropy(B, C, ld):
    """dataset/tud.py:25-27"""
    from graphtrans_amd import losses
    g = torch.Generator().manual_seed(3)
    buf = torch.randn(B, ld, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    ref = buf[:, :C].clone().requires_grad_()
    want = torch.nn.functional.cross_entropy(ref, y)
    want.backward()
    pred = buf.cuda()[:, :C].requires_grad_()
    got = losses.tud_loss(pred, y.cuda())
    got.backward()
    assert abs(got.item() - want.item()) <= 1e-5
    assert (pred.grad.cpu() - ref.grad).abs().max().item() <= 1e-6
    with pytest.raises(RuntimeError):
        losses.tud_loss(ref, y)


def test_xent_reports_out_of_range_targets_in_validate_mode():
    """torch.nn.CrossEntropyLoss (dataset/code.py:42) asserts on a class index outside [0, C); the kernel counts such
    targets into a status word and the wrapper raises when validating (off by default: the read is a device sync)."""
    from graphtrans_amd import ops
    torch.manual_seed(0)
    x = torch.randn(6, 3, 11, device=DEV, requires_grad=True)
    t = torch.randint(0, 11, (6, 3), device=DEV)
    t[1, 2] = -100          # ignore_index: fine
    ops.set_validate(True)
    try:
        ops.softmax_xent(x, t).backward()
        t[4, 0] = 11
        t[0, 1] = -3
        with pytest.raises(IndexError, match="2 target"):
            ops.softmax_xent(x, t)
    finally:
        ops.set_validate(False)
    assert torch.isfinite(ops.softmax_xent(x, t))   # not validating: treated as ignored rows, as before


def test_standalone_dropout_replays_its_mask():
    """gt_dropout (nn.Dropout of the masked encoder / F.dropout of PNANodeEmbedding): keep rate, scaling, and the backward
    applying the same mask."""
    from graphtrans_amd import ops
    torch.manual_seed(0)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(4096, 64, device=DEV).to(dt).requires_grad_(True)
        y = ops.dropout(x, 0.25, True, seed=77)
        keep = y != 0
        assert abs(keep.float().mean().item() - 0.75) < 0.01
        assert torch.allclose(y[keep].float(), x.detach()[keep].float() / 0.75, rtol=1e-2 if dt == torch.bfloat16 else 1e-6)
        g = torch.randn_like(y)
        y.backward(g)
        assert torch.equal(x.grad != 0, keep & (g != 0))
        assert torch.allclose(x.grad[keep].float(), g[keep].float() / 0.75, rtol=1e-2 if dt == torch.bfloat16 else 1e-6)
        assert torch.equal(ops.dropout(x, 0.25, True, seed=77), y) and ops.dropout(x, 0.25, False) is x
