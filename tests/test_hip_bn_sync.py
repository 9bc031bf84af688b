"""The statistics-exchange protocol of the library's BatchNorm (include/graphtrans_hip.h: gt_bn_sync_set) without a process group:
a hook that plays `world` IDENTICAL ranks (all-gather = the local packet repeated, all-reduce = the local sums x world) must leave
every result unchanged -- merged mean / variance of identical shards are the shard's own, and (sums, count) scale together."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class FakeRanks:
    def __init__(self, world):
        from graphtrans_amd import _lib
        from graphtrans_amd.dist import _DevView
        self.world, self.calls, self._view = world, [0, 0], _DevView
        self._cb = _lib.BN_SYNC_FN(self._hook)

    def _hook(self, user, kind, buf, n, stream):
        try:
            n = int(n)
            # (NULL = torch's default stream; ExternalStream(0) would be a fresh pool stream)
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream)) if stream else torch.cuda.default_stream()):
                if kind == 0:
                    v = torch.as_tensor(self._view(buf, n * (self.world + 1)), device=DEV)
                    v[n:].copy_(v[:n].repeat(self.world))
                else:
                    v = torch.as_tensor(self._view(buf, n), device=DEV)
                    v.mul_(float(self.world))
            self.calls[kind] += 1
            return 0
        except Exception as e:  # pragma: no cover
            print("hook failed:", repr(e))
            return -1

    def __enter__(self):
        from graphtrans_amd import _lib
        _lib.check(_lib.lib().gt_bn_sync_set(C.cast(self._cb, C.c_void_p), None, self.world), "gt_bn_sync_set")
        return self

    def __exit__(self, *a):
        from graphtrans_amd import _lib
        _lib.lib().gt_bn_sync_set(None, None, 1)


@pytest.mark.parametrize("rows,D", [(5000, 300), (256, 600), (7, 64), (1500, 128)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_op_is_unchanged_by_identical_ranks(rows, D, relu):
    from graphtrans_amd import ops
    torch.manual_seed(rows + D)
    x = torch.randn(rows, D, device=DEV) * 2 + 0.5
    w, b = torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV)
    g = torch.randn(rows, D, device=DEV)

    def run(hooked):
        xx, ww, bb = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rm, rv, nbt = torch.zeros(D, device=DEV), torch.ones(D, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
        y = ops.batch_norm(xx, ww, bb, rm, rv, nbt, 0.1, 1e-5, True, relu)
        if hooked is not None:   # (the backward of a plain autograd op runs on autograd's thread: call the C entry points here instead)
            pass
        y.backward(g)
        torch.cuda.synchronize()
        return [t.detach().clone() for t in (y, xx.grad, ww.grad, bb.grad, rm, rv)]

    ref = run(None)
    # forward under the hook (this thread); the backward thread has no hook -> compare the forward side only here
    with FakeRanks(3) as fk:
        xx = x.clone()
        rm, rv, nbt = torch.zeros(D, device=DEV), torch.ones(D, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
        with torch.no_grad():
            y = ops.batch_norm(xx, w, b, rm, rv, nbt, 0.1, 1e-5, True, relu)
        torch.cuda.synchronize()
        assert fk.calls[0] == 1
    assert torch.allclose(y, ref[0], rtol=1e-5, atol=1e-5), float((y - ref[0]).abs().max())
    assert torch.allclose(rm, ref[4], rtol=1e-5, atol=1e-6)
    # running_var: unbiased with the GLOBAL count (3 x rows) instead of rows
    n = rows
    want = 0.9 + 0.1 * (ref[5] - 0.9) / 0.1 * ((n - 1) / n) * (3 * n / (3 * n - 1))
    assert torch.allclose(rv, want, rtol=1e-4, atol=1e-5)
    assert int(nbt) == 1


@pytest.mark.parametrize("gnn_type", ["gcn", "gin"])
def test_fused_model_is_unchanged_by_identical_ranks(gnn_type, monkeypatch):
    """whole fused forward + backward (every BatchNorm of the composites, main and virtual-node streams) under the fake-ranks hook,
    installed on both threads through the engine's own hook slot"""
    import copy

    from test_hip_dp import _model, _shard
    from graphtrans_amd import engine, losses
    model = _model(gnn_type)
    ref_model = copy.deepcopy(model)
    b, y = _shard(0)
    losses.code2_loss(ref_model(b), y).backward()
    fk = FakeRanks(2)

    class Slot:   # what engine expects of a hook object
        error = None

        def install(self):
            fk.__enter__()

        def uninstall(self):
            fk.__exit__()
    monkeypatch.setattr(engine, "_bn_sync_hook", lambda m, p: Slot())
    assert engine.eligible(model, b, None)
    losses.code2_loss(model(b), y).backward()
    torch.cuda.synchronize()
    assert fk.calls[0] > 0 and fk.calls[1] > 0
    for (n, p), q in zip(model.named_parameters(), ref_model.parameters()):
        scale = max(1e-3, float(q.grad.abs().max()))
        assert float((p.grad - q.grad).abs().max()) <= 2e-4 * scale, (n, float((p.grad - q.grad).abs().max()), scale)
