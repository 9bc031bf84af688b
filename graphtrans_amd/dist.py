"""Data parallelism by graph sharding: one process per GPU, gradients averaged with a bucketed
all-reduce over RCCL (torch.distributed backend "nccl" on ROCm) overlapped with backward.

The reference has no parallelism at all (main.py:109,174); this is new.  Graphs
never exchange messages (edge_index is block diagonal, attention is per graph), so the ONLY
collective on the path is the gradient all-reduce.  Parameters (9.05 M for the Code2 config,
36 MB fp32) are replicated; gradients live in a few flat buckets so that
  * gradients are never zeroed or accumulated (set to None, then assigned by autograd),
  * after backward each bucket is packed with one multi-tensor copy and reduced asynchronously on
    RCCL's own stream while the next bucket is being packed,
  * xGMI is a point-to-point mesh: few, large messages (default 16 MB buckets) keep every link busy.
BatchNorm statistics stay per rank (what torch DDP does); see DESIGN.md for the consequences.

Fused model path (engine.py): all gradients already live in ONE flat buffer, so no packing is needed
and the reduction is issued from inside the backward: the transformer + heads half of the buffer goes
on the wire while the message-passing backward is still running (`attach` / `reduce_flat`).
"""
import torch
import torch.distributed as dist


class GradSync:
    """zero() / finish() around backward.

    world == 1: gradients are simply dropped to None before backward (autograd then ASSIGNS each
    produced gradient instead of launching one accumulate kernel per parameter, ~140 launches per
    step) and finish() is a no-op.
    world  > 1: same, plus after backward the gradients are packed into flat buckets with one
    multi-tensor copy per bucket, all-reduced asynchronously (bucket k+1 is packed while bucket k
    is on the wire), averaged, and `.grad` is re-pointed at the reduced bucket views.
    """

    def __init__(self, params, world_size=None, bucket_bytes=16 << 20, group=None, always_reduce=False):
        self.group = group
        self.always_reduce = always_reduce   # issue the collective even for one rank (tests)
        self._pending, self._flat_used = [], False
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []  # dict(flat, params, views)
        cur, cur_bytes = [], 0
        for p in reversed(self.params):  # reverse registration order = the order backward produces them
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._make_bucket(cur)

    def _make_bucket(self, ps):
        total = sum(p.numel() for p in ps)
        flat = torch.zeros(total, dtype=ps[0].dtype, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.buckets.append(dict(flat=flat, params=ps, views=views))

    def attach(self, model):
        """Let the fused model path (engine.py) hand its flat gradient buffer to this reducer."""
        from . import engine
        engine.state(model)["sync"] = self
        return self

    @property
    def active(self):
        """reduce_flat() puts data on the wire (the fused backward joins its weight-gradient stream before it only then)"""
        return self.world > 1 or self.always_reduce

    def reduce_flat(self, flat, lo, hi):
        """Average flat[lo:hi] over the ranks, asynchronously (called from the fused backward as soon as
        the kernels producing that range are enqueued).  finish() waits for it."""
        if self.world == 1 and not self.always_reduce:
            return
        seg = flat[lo:hi]
        if self.world > 1:
            seg.div_(self.world)
        self._pending.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._flat_used = True

    def zero(self):
        """Replaces optimizer.zero_grad(set_to_none=True)."""
        for p in self.params:
            p.grad = None

    def finish(self):
        if self._flat_used:   # the fused backward already issued the reductions on its flat buffer
            for h in self._pending:
                h.wait()
            self._pending, self._flat_used = [], False
            return
        if self.world == 1:
            return
        handles = []
        for b in self.buckets:
            have = [(v, p.grad) for v, p in zip(b["views"], b["params"]) if p.grad is not None]
            if len(have) != len(b["params"]):
                b["flat"].zero_()  # parameters unused this step contribute zeros
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            b["flat"].div_(self.world)  # average (gloo has no AVG op; pre-scaling keeps fp32 range)
            handles.append(dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for b, h in zip(self.buckets, handles):
            h.wait()
            for v, p in zip(b["views"], b["params"]):
                p.grad = v

    def grad_bytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)


def balanced_shards(sizes, world, cost=None):
    """Split one global batch of graphs over `world` ranks with equal graph counts (+-1) and balanced work.

    sizes: nodes per graph (host array).  cost(n) defaults to n^2 + 64 n: attention is quadratic in the graph
    size, message passing and the GEMMs linear (SURVEY.md 8e "size-balanced assignment").  Greedy longest-
    processing-time: graphs in decreasing cost order go to the least-loaded rank that still has room.
    Returns a list of `world` int64 index arrays (each in increasing order, so that collation order within a
    shard follows the sampler's order).  Deterministic; every rank computes the same split from the same ids."""
    import numpy as np
    sizes = np.asarray(sizes, dtype=np.int64)
    n = sizes.size
    c = (sizes.astype(np.float64) ** 2 + 64.0 * sizes) if cost is None else np.asarray([cost(int(s)) for s in sizes], np.float64)
    cap = [n // world + (1 if r < n % world else 0) for r in range(world)]
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in np.argsort(-c, kind="stable"):
        r = min((r for r in range(world) if len(out[r]) < cap[r]), key=lambda r: (load[r], r))
        out[r].append(int(i))
        load[r] += float(c[i])
    return [np.array(sorted(o), dtype=np.int64) for o in out]
