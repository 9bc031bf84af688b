"""N>1 data-parallel path on CPU: world_size-2 gloo run of graphtrans_amd.dist.GradSync.
Property: with equal-size shards and a mean loss, the averaged gradients equal the single-process
full-batch gradients; parameters without a gradient (unused branch) do not dead-lock."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graphtrans_amd.dist import GradSync


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16),
                                        torch.nn.ReLU(), torch.nn.Linear(16, 3))
        self.unused = torch.nn.Linear(4, 4)  # never called: its bucket must still be reduced

    def forward(self, x):
        return self.body(x)


def _model():
    torch.manual_seed(0)
    return _Net()


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model()
    sync = GradSync(m.parameters(), bucket_bytes=512)  # tiny buckets -> several of them
    assert len(sync.buckets) > 2
    x, y = _data()
    xs, ys = x[rank::world], y[rank::world]
    for _ in range(2):  # second iteration checks zero() re-arms the buckets
        sync.zero()
        ((m(xs) - ys) ** 2).mean().backward()
        sync.finish()
    out[rank] = [p.grad.clone() for p in m.parameters()]  # unused params hold the (zero) reduced bucket view
    dist.destroy_process_group()


def test_gradsync_world2_matches_full_batch():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    m = _model()
    x, y = _data()
    ((m(x) - y) ** 2).mean().backward()
    ref = [p.grad if p.grad is not None else torch.zeros_like(p) for p in m.parameters()]
    for r in (0, 1):
        for a, b in zip(out[r], ref):
            assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()


def test_gradsync_single_process_views():
    m = _model()
    sync = GradSync(m.parameters(), world_size=1)
    x, y = _data()
    sync.zero()
    ((m(x) - y) ** 2).mean().backward()
    sync.finish()
    assert all(p.grad is not None for p in m.body.parameters()) and all(p.grad is None for p in m.unused.parameters())
    sync.zero()
    assert all(p.grad is None for p in m.parameters())


def _flat_worker(rank, world, port, out):
    """The fused model path's protocol: gradients are views of one flat buffer, reduce_flat() is called
    per finished range from inside the backward, finish() waits."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model()
    params = list(m.body.parameters())
    sync = GradSync(params)
    x, y = _data()
    xs, ys = x[rank::world], y[rank::world]
    total = sum(p.numel() for p in params)
    flat = torch.zeros(total)
    sync.zero()
    grads = torch.autograd.grad(((m(xs) - ys) ** 2).mean(), params)
    off, views = 0, []
    for p, g in zip(params, grads):
        v = flat[off:off + p.numel()].view_as(p)
        v.copy_(g)
        p.grad = v
        views.append(v)
        off += p.numel()
    half = total // 2
    sync.reduce_flat(flat, half, total)
    sync.reduce_flat(flat, 0, half)
    sync.finish()
    assert not sync._pending and not sync._flat_used
    out[rank] = [p.grad.clone() for p in params]
    dist.destroy_process_group()


def test_gradsync_flat_buffer_protocol_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_flat_worker, args=(2, port, out), nprocs=2, join=True)
    m = _model()
    x, y = _data()
    ((m(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in m.body.parameters()]
    for r in (0, 1):
        for a, b in zip(out[r], ref):
            assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
