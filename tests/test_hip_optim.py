"""FusedAdamW (gt_adamw_step) against torch.optim.AdamW (the reference's optimizer, main.py:178)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _models():
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(37, 301), torch.nn.ReLU(), torch.nn.Linear(301, 5), torch.nn.Embedding(11, 3)).to(DEV)
    b = torch.nn.Sequential(torch.nn.Linear(37, 301), torch.nn.ReLU(), torch.nn.Linear(301, 5), torch.nn.Embedding(11, 3)).to(DEV)
    b.load_state_dict(a.state_dict())
    return a, b


def _loss(m, x, use_emb):
    y = m[2](m[1](m[0](x))).square().mean()
    if use_emb:
        y = y + m[3].weight.sum() * 0.1
    return y


@pytest.mark.parametrize("wd", [0.0, 0.05])
def test_fused_adamw_matches_torch(wd):
    from graphtrans_amd.optim import FusedAdamW
    a, b = _models()
    oa = torch.optim.AdamW(a.parameters(), lr=3e-3, weight_decay=wd, betas=(0.9, 0.99), eps=1e-8)
    ob = FusedAdamW(b.parameters(), lr=3e-3, weight_decay=wd, betas=(0.9, 0.99), eps=1e-8)
    sched = torch.optim.lr_scheduler.StepLR(ob, step_size=3, gamma=0.5)
    scheda = torch.optim.lr_scheduler.StepLR(oa, step_size=3, gamma=0.5)
    for i in range(8):
        x = torch.randn(64, 37, device=DEV)
        use_emb = i % 3 != 1  # the embedding has no gradient on some steps: torch skips it (its step count lags)
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad(set_to_none=True)
            _loss(m, x, use_emb).backward()
            o.step()
        sched.step()
        scheda.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7), (pa - pb).abs().max()
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for k in sa:
        assert float(sa[k]["step"]) == float(sb[k]["step"])
        assert torch.allclose(sa[k]["exp_avg"], sb[k]["exp_avg"], rtol=1e-5, atol=1e-8)
        assert torch.allclose(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"], rtol=1e-5, atol=1e-10)


def test_fused_adamw_state_dict_roundtrip():
    from graphtrans_amd.optim import FusedAdamW
    a, b = _models()
    oa = FusedAdamW(a.parameters(), lr=1e-2)
    x = torch.randn(16, 37, device=DEV)
    for _ in range(3):
        oa.zero_grad(set_to_none=True)
        _loss(a, x, True).backward()
        oa.step()
    b.load_state_dict(a.state_dict())
    ob = FusedAdamW(b.parameters(), lr=1e-2)
    ob.load_state_dict(oa.state_dict())
    for m, o in ((a, oa), (b, ob)):
        o.zero_grad(set_to_none=True)
        _loss(m, x, True).backward()
        o.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)


def test_fused_adamw_rejects_cpu_parameters():
    from graphtrans_amd.optim import FusedAdamW
    m = torch.nn.Linear(4, 4)
    o = FusedAdamW(m.parameters())
    m(torch.randn(2, 4)).sum().backward()
    with pytest.raises(RuntimeError):
        o.step()


@pytest.mark.parametrize("max_norm", [0.05, 1e3])
def test_fused_adamw_grad_clip_matches_clip_grad_norm(max_norm):
    """trainers/base_trainer.py:34-36: clip_grad_norm_(model.parameters(), grad_clip) then optimizer.step();
    0.05 clips on every step here, 1e3 never does (coefficient clamped to 1)."""
    from graphtrans_amd.optim import FusedAdamW
    a, b = _models()
    oa = torch.optim.AdamW(a.parameters(), lr=3e-3, weight_decay=0.01)
    ob = FusedAdamW(b.parameters(), lr=3e-3, weight_decay=0.01, max_grad_norm=max_norm)
    for i in range(5):
        x = torch.randn(64, 37, device=DEV)
        use_emb = i != 2
        oa.zero_grad(set_to_none=True)
        _loss(a, x, use_emb).backward()
        norm = torch.nn.utils.clip_grad_norm_(a.parameters(), max_norm)
        oa.step()
        ob.zero_grad(set_to_none=True)
        _loss(b, x, use_emb).backward()
        ob.step()
        assert torch.allclose(ob.last_grad_norm, norm, rtol=1e-5), (ob.last_grad_norm, norm)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7), (pa - pb).abs().max()
