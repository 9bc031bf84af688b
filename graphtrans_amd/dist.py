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
import contextlib
import ctypes as C

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
        self.stats = dict(collectives=0, bytes=0)   # gradient collectives issued and bytes put on the wire by this rank (bench.py)
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
        self.stats["collectives"] += 1
        self.stats["bytes"] += seg.numel() * seg.element_size()
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
            self.stats["collectives"] += 1
            self.stats["bytes"] += b["flat"].numel() * b["flat"].element_size()
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


# ---- synchronised BatchNorm statistics for the library's BatchNorm calls (composite / fused paths) -----------------------
class _DevView:
    """n floats at a raw device pointer as a torch tensor (no copy): torch.as_tensor reads __cuda_array_interface__"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class BnSyncHook:
    """The collective half of synchronised BatchNorm (include/graphtrans_hip.h: gt_bn_sync_set).  While installed for the
    calling thread, every training-mode BatchNorm inside the C library -- the ones in gt_gcn_layer_* / gt_gin_layer_* /
    gt_vn_update_* included, i.e. the FUSED model path -- exchanges (rows, mean, var) by all-gather in its forward and
    (sum dy', sum dy' xhat, rows) by all-reduce in its backward, on the stream it runs on.  With graphs sharded over ranks
    the model then is the reference's single-device-batch model (modules/gnn_module.py:204,164,167).

        hook = BnSyncHook(group)            # once
        with hook.installed(): ...          # per thread (autograd's backward thread installs it again)
    RCCL ("nccl") moves device buffers on the BatchNorm's stream; other backends (gloo in the tests) go through the host."""

    def __init__(self, group=None):
        import torch.distributed as dist
        from . import _lib
        self.group = group
        self.world = dist.get_world_size(group)
        self.device_collectives = dist.get_backend(group) == "nccl"
        self._streams = {}
        self.calls = 0
        self.error = None
        self._cb = _lib.BN_SYNC_FN(self._hook)   # keep the ctypes thunk alive as long as the hook

    def _stream(self, ptr, device):
        """torch stream object of the raw stream the BatchNorm runs on.  NULL is torch's default stream: ExternalStream(0) is NOT
        -- torch.cuda.Stream(stream_ptr=0) takes a fresh stream from torch's pool, and a copy issued there overtakes the
        kernels on the NULL stream (found as stale statistics packets in one exchange out of ~10)."""
        if not ptr:
            return torch.cuda.default_stream(device)
        st = self._streams.get(ptr)
        if st is None:
            st = self._streams[ptr] = torch.cuda.ExternalStream(int(ptr), device=device)
        return st

    def _hook(self, user, kind, buf, n, stream):
        import torch.distributed as dist
        try:
            dev = torch.device("cuda", torch.cuda.current_device())
            n, w = int(n), self.world
            with torch.cuda.stream(self._stream(stream, dev)):
                if kind == 0:
                    view = torch.as_tensor(_DevView(buf, n * (w + 1)), device=dev)
                    local, gathered = view[:n], view[n:]
                    if self.device_collectives:
                        dist.all_gather_into_tensor(gathered, local, group=self.group)
                    else:
                        parts = [torch.empty(n, dtype=torch.float32) for _ in range(w)]
                        dist.all_gather(parts, local.cpu(), group=self.group)
                        gathered.copy_(torch.cat(parts))
                else:
                    view = torch.as_tensor(_DevView(buf, n), device=dev)
                    if self.device_collectives:
                        dist.all_reduce(view, group=self.group)
                    else:
                        host = view.cpu()
                        dist.all_reduce(host, group=self.group)
                        view.copy_(host)
            self.calls += 1
            return 0
        except Exception as e:   # never unwind through the C frames
            self.error = e
            return -1

    def install(self):
        from . import _lib
        _lib.check(_lib.lib().gt_bn_sync_set(C.cast(self._cb, C.c_void_p), None, self.world), "gt_bn_sync_set")

    @staticmethod
    def uninstall():
        from . import _lib
        _lib.lib().gt_bn_sync_set(None, None, 1)

    @contextlib.contextmanager
    def installed(self):
        self.install()
        try:
            yield self
        finally:
            self.uninstall()
            if self.error is not None:
                e, self.error = self.error, None
                raise e
