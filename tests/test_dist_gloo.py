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


def _mol_data():
    g = torch.Generator().manual_seed(2)
    pred = torch.randn(8, 5, generator=g)
    y = (torch.rand(8, 5, generator=g) > 0.5).float()
    y[torch.rand(8, 5, generator=g) < 0.5] = float("nan")
    y[0::2, :3] = float("nan")   # rank 0's shard (rows 0, 2, 4, 6) holds far fewer labels than rank 1's
    return pred, y


def _mol_worker(rank, world, port, out):
    from graphtrans_amd import losses
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pred, y = _mol_data()
    p, ys = pred[rank::world].clone().requires_grad_(), y[rank::world]
    den = losses.dp_label_denominator(ys)
    # what gt_bce_masked_fwd/bwd compute with den_in = den (csrc/xent.hip), stated with torch on the CPU
    m = ys == ys
    per = torch.nn.functional.binary_cross_entropy_with_logits(p, torch.where(m, ys, torch.zeros(())), reduction="none")
    loss = (per * m).sum() / den
    loss.backward()
    out[rank] = (float(den), loss.item(), p.grad.clone())
    dist.destroy_process_group()


def test_molpcba_loss_denominator_world2_is_the_global_mean():
    """the Molpcba loss normalises by the labelled-entry count (dataset/mol.py:25-27); with shards
    of different counts the rank losses use global count / world so that the AVERAGED gradient is exact."""
    from oracle import reference_math as rm
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_mol_worker, args=(2, port, out), nprocs=2, join=True)
    pred, y = _mol_data()
    pr = pred.clone().requires_grad_()
    want = rm.mol_loss(pr, y)
    want.backward()
    n = float((y == y).sum())
    assert out[0][0] == out[1][0] == n / 2
    assert abs((out[0][1] + out[1][1]) / 2 - want.item()) < 1e-6          # mean of the rank losses = global loss
    for r in (0, 1):                                                       # rank-averaged gradient = global gradient
        assert torch.allclose(out[r][2] / 2, pr.grad[r::2], atol=1e-7)


def test_balanced_shards():
    import numpy as np
    from graphtrans_amd import synth
    from graphtrans_amd.dist import balanced_shards
    sizes = np.bincount(synth.code2_like(B=256, seed=0).batch.numpy(), minlength=256)
    for world in (1, 2, 4, 8):
        shards = balanced_shards(sizes, world)
        assert sorted(np.concatenate(shards).tolist()) == list(range(256))          # a partition
        assert {len(s) for s in shards} == {256 // world}                           # equal graph counts
        cost = [float((sizes[s].astype(np.float64) ** 2 + 64.0 * sizes[s]).sum()) for s in shards]
        naive = [float((sizes[r::world].astype(np.float64) ** 2 + 64.0 * sizes[r::world]).sum()) for r in range(world)]
        assert max(cost) <= max(naive) + 1e-9
        if world > 1:
            assert max(cost) / (sum(cost) / world) < 1.05, cost                     # within 5 % of perfect balance
    # uneven division: counts differ by at most one
    assert sorted(len(s) for s in balanced_shards(sizes[:10], 4)) == [2, 2, 3, 3]


def test_bench_self_spawns_its_ranks():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run with 2 ranks
    (bench.py:self_spawn); --dry-run keeps the GPU out of it: rendezvous on 127.0.0.1, barrier, MAX over ranks, ONE
    JSON line from rank 0 with n_gpus = 2."""
    import json
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["max_over_ranks"] == 2.0


def _ddp_probe(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphtrans_amd import engine
    m = torch.nn.Sequential(torch.nn.Linear(4, 4))
    res = [engine.wrapped_in_ddp(m), engine.has_grad_hooks(m)]
    wrapper = torch.nn.parallel.DistributedDataParallel(m)
    res.append(engine.wrapped_in_ddp(m))
    holder = torch.nn.ModuleDict({"module": torch.nn.Linear(2, 2)})   # a plain container with a child called "module" is not DDP
    res.append(engine.wrapped_in_ddp(holder["module"]))
    m2 = torch.nn.Linear(3, 3)
    m2.weight.register_hook(lambda g: g)
    m3 = torch.nn.Linear(3, 3)
    m3.bias.register_post_accumulate_grad_hook(lambda p: None)
    res += [engine.has_grad_hooks(m2), engine.has_grad_hooks(m3)]
    out.put(res)
    del wrapper
    dist.destroy_process_group()


def test_fused_path_declines_models_under_ddp_or_with_gradient_hooks():
    """The fused autograd node assigns `.grad` itself: DistributedDataParallel's reducer hooks and user tensor hooks would never
    fire, so engine.eligible sends such models through the module-by-module path."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_ddp_probe, args=(0, 1, port, out))
    p.start()
    res = out.get(timeout=120)
    p.join(60)
    assert res == [False, False, True, False, True, True]
