"""Per-batch index structures for the HIP kernels (built once per collated batch on device).

Replaces the reference's per-layer `degree(row)` (modules/conv.py:57), its unsorted scatter
(conv.py:28,63) and the `batch.eq(i)` loops of pad_batch (modules/utils.py:9-13) with one call to
`gt_graph_prep`, plus the sequence layouts (`SeqLayout`) that describe how node rows map to
transformer token rows (padded = the reference's (S,B,d) layout; packed = no padding rows at all).
"""
import threading

import numpy as np
import torch

from . import _lib


def _stream():
    """raw hipStream_t of torch's current stream.  With an EXPLICIT device index: torch.cuda.current_stream() without one
    resolves "the current device" through torch._utils._get_available_device_type() -> torch.cuda.is_available() ->
    hipGetDeviceCount, 110 us of host time per call on the MI355X boxes (rocprofv3 --hip-trace: 5 calls = 0.55 ms per step)."""
    return torch.cuda.current_stream(torch.cuda.current_device()).cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


class GraphStructure:
    """graph_ptr, node_graph, CSR by destination (in_*), CSC by source (out_*), deg, dis."""

    __slots__ = ("N", "E", "B", "device", "graph_ptr", "node_graph", "in_ptr", "in_src", "in_eid", "out_ptr",
                 "out_dst", "out_eid", "deg", "dis", "status", "_sizes", "_layouts", "_pna_scales", "_ws", "ready_event")

    @staticmethod
    def build(edge_index, batch, num_graphs=None, sizes=None, stream=None):
        """edge_index (2,E) int64, batch (N,) int64 sorted; both on the GPU.  `num_graphs` / `sizes`
        (host values, e.g. PyG Batch.num_graphs / the collater's per-graph node counts) avoid the
        device sync the reference performs at modules/gnn_module.py:195."""
        if not edge_index.is_cuda:
            raise RuntimeError("graphtrans_amd kernels run on the GPU only (no CPU fallback)")
        gs = GraphStructure()
        dev = edge_index.device
        N, E = int(batch.numel()), int(edge_index.shape[1])
        if num_graphs is None:
            num_graphs = len(sizes) if sizes is not None else (int(batch[-1].item()) + 1 if N > 0 else 0)
        B = int(num_graphs)
        edge_index = edge_index.contiguous()
        batch = batch.contiguous()
        if edge_index.dtype != torch.int64 or batch.dtype != torch.int64:
            raise TypeError("edge_index and batch must be int64 (PyG collation dtype)")
        i32 = dict(dtype=torch.int32, device=dev)
        gs.N, gs.E, gs.B, gs.device = N, E, B, dev
        gs.graph_ptr = torch.empty(B + 1, **i32)
        gs.node_graph = torch.empty(max(N, 1), **i32)
        gs.in_ptr = torch.empty(N + 1, **i32)
        gs.out_ptr = torch.empty(N + 1, **i32)
        idx = torch.empty(4, max(E, 1), **i32)
        gs.in_src, gs.in_eid, gs.out_dst, gs.out_eid = idx[0], idx[1], idx[2], idx[3]
        dd = torch.empty(2, max(N, 1), dtype=torch.float32, device=dev)
        gs.deg, gs.dis = dd[0], dd[1]
        gs.status = torch.empty(1, **i32)
        L = _lib.lib()
        ws_bytes = L.gt_graph_prep_workspace_bytes(N, E, B)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.launch("gt_graph_prep", _ptr(edge_index), _ptr(batch), N, E, B, _ptr(gs.graph_ptr), _ptr(gs.node_graph),
                    _ptr(gs.in_ptr), _ptr(gs.in_src), _ptr(gs.in_eid), _ptr(gs.out_ptr), _ptr(gs.out_dst),
                    _ptr(gs.out_eid), _ptr(gs.deg), _ptr(gs.dis), _ptr(gs.status), _ptr(ws), ws_bytes, _stream() if stream is None else stream)
        gs._ws = ws   # (with a caller's stream the workspace must outlive this call's allocator scope)
        gs.ready_event = None   # set by a caller that built on a side stream: consumers wait for it once (engine.prep_*)
        gs._sizes = None if sizes is None else np.asarray(sizes, dtype=np.int64)
        gs._layouts = {}
        gs._pna_scales = None
        return gs

    def validate(self):
        """Raise if gt_graph_prep saw an out-of-range edge/batch index (device sync)."""
        s = int(self.status.item())
        if s:
            raise ValueError("edge_index / batch out of range or batch not sorted (gt_graph_prep status %d)" % s)

    @property
    def sizes(self):
        """Host per-graph node counts (one small D2H copy if the collater did not supply them)."""
        if self._sizes is None:
            p = self.graph_ptr.cpu().numpy().astype(np.int64)
            self._sizes = np.diff(p)
        return self._sizes

    def layout(self, kind, max_input_len, with_cls):
        key = (kind, int(max_input_len), bool(with_cls))
        if key not in self._layouts:
            if kind == "packed" and self._sizes is None and torch.device(self.device).type == "cuda":
                # sizes unknown on the host (a bare device batch): build the layout on the device, no D2H sync
                self._layouts[key] = SeqLayout.packed_on_device(self, int(max_input_len), bool(with_cls))
            else:
                self._layouts[key] = SeqLayout(self, kind, int(max_input_len), bool(with_cls))
        return self._layouts[key]


class _StageRing:
    """ONE pinned host allocation cut into slots (pinning costs ~1 ms per allocation), handed out round robin; a slot is reused
    only after the H2D copy that last read it has completed (waited for if the host is more than a ring ahead of the device)."""
    SLOTS, SLOT_BYTES = 64, 1 << 16

    def __init__(self):
        self.buf = torch.empty(self.SLOTS * self.SLOT_BYTES, dtype=torch.uint8).pin_memory()
        self.events = [None] * self.SLOTS
        self.next = 0

    def take(self, nbytes):
        if nbytes > self.SLOT_BYTES:
            return None, None
        i = self.next
        self.next = (i + 1) % self.SLOTS
        if self.events[i] is None:
            self.events[i] = torch.cuda.Event()
        else:
            self.events[i].synchronize()
        return self.buf[i * self.SLOT_BYTES:(i + 1) * self.SLOT_BYTES], self.events[i]


_RINGS = {}   # one ring per device: its events belong to that device's streams (one process-global ring reused events across devices)
_RINGS_LOCK = threading.Lock()


def _staging(nbytes, device):
    """(pinned slot, event to record behind the copy that reads it) from the ring of `device`; (None, None) if it does not fit a slot"""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    with _RINGS_LOCK:   # (autograd's backward threads and data-loader threads build layouts too)
        ring = _RINGS.get(key)
        if ring is None:
            ring = _RINGS[key] = _StageRing()
        with torch.cuda.device(key):   # a new slot's event is created on the ring's device
            return ring.take(nbytes)


class SeqLayout:
    """seq_desc[B][4] = {row0, npos, kv_off, kv_len} (see include/graphtrans_hip.h).

    exact: rows / num_work / max_npos are the true counts (host-built).  A device-built layout (packed_on_device) only
    knows upper bounds on the host: its token buffers must be ZERO-initialised, the rows past the true count then stay
    finite through every row-wise kernel and contribute exactly 0 to every weight gradient.

    padded: the reference layout of pad_batch + CLS (modules/utils.py:5-29,
            modules/transformer_encoder.py:50-55): rows = S' x B, position-major, left padded.
    packed: only real tokens, graph after graph: rows = sum_b (kept_b + cls).
    """

    exact = True

    @classmethod
    def packed_on_device(cls, gs, max_input_len, with_cls):
        """gt_seq_layout_packed: desc / last_rows / work list from graph_ptr on the device (the reference's pad_batch loop,
        modules/utils.py:9-16, costs O(B) device syncs; the host-built layout one D2H copy when sizes are unknown)."""
        self = cls.__new__(cls)
        B, N, c = gs.B, gs.N, 1 if with_cls else 0
        dev = gs.device
        self.kind, self.with_cls, self.B, self.exact = "packed", with_cls, B, False
        self.row_stride = 1
        self.rows = N + B * c                      # upper bound: truncation (n > max_input_len) only removes rows
        self.max_npos = min(int(max_input_len), N) + c
        self.num_work = B + self.rows // 64        # upper bound on sum_b ceil(kv_len_b / 64)
        self.desc = torch.empty((B, 4), dtype=torch.int32, device=dev)
        self.last_rows = torch.empty(B, dtype=torch.int64, device=dev)
        self.work = torch.empty((max(self.num_work, 1), 2), dtype=torch.int32, device=dev)
        self.meta = torch.empty(4, dtype=torch.int32, device=dev)
        _lib.launch("gt_seq_layout_packed", _ptr(gs.graph_ptr), B, int(max_input_len), c, _ptr(self.desc), _ptr(self.last_rows),
                    _ptr(self.work), self.num_work, _ptr(self.meta), _stream())
        return self

    def __init__(self, gs, kind, max_input_len, with_cls):
        n = gs.sizes
        B = gs.B
        cls = 1 if with_cls else 0
        S = int(min(int(n.max()) if B else 0, max_input_len))  # modules/utils.py:16
        kept = np.minimum(n, S)
        kv_len = kept + cls
        desc = np.zeros((B, 4), dtype=np.int32)
        if kind == "padded":
            npos = S + cls
            desc[:, 0] = np.arange(B)
            desc[:, 1] = npos
            desc[:, 2] = npos - kv_len
            desc[:, 3] = kv_len
            self.row_stride, self.rows, self.max_npos = B, npos * B, npos
        elif kind == "packed":
            tok_ptr = np.concatenate([[0], np.cumsum(kv_len)])
            desc[:, 0] = tok_ptr[:-1]
            desc[:, 1] = kv_len
            desc[:, 2] = 0
            desc[:, 3] = kv_len
            self.row_stride, self.rows, self.max_npos = 1, int(tok_ptr[-1]), int(kv_len.max()) if B else 0
            self.tok_ptr = tok_ptr
        else:
            raise ValueError(kind)
        self.kind, self.with_cls, self.S, self.B = kind, with_cls, S, B
        self.kept = kept
        self.desc_cpu = desc
        # attention tile list: (sequence, 64-position tile) for every tile that exists (ragged sizes)
        # Longest sequences FIRST: a block walks all keys (queries) of its sequence tile by tile, so the longest sequence's
        # blocks are a serial chain several times the typical one (Code2-like: 418 against 126 tokens) -- started last it
        # was the tail of every attention launch.  The kernels give XCD x the x-th contiguous eighth of the list (one L2 per
        # sequence) and dispatch each eighth front to back: sequences are dealt to the eighths by length rank, each eighth
        # holds its own in descending length, padded with {-1, 0} entries (skipped) to equal size.
        tiles = (desc[:, 1].astype(np.int64) + 63) // 64
        if B:
            order = np.argsort(-desc[:, 1].astype(np.int64), kind="stable")
            xcd = np.arange(B) % 8                              # eighth of the sequence of length rank r
            perm = order[np.argsort(xcd, kind="stable")]        # = concat(order[x::8] for x in 0..7)
            xs = np.sort(xcd)                                   # eighth of perm[i]
            t = tiles[perm]
            cnt = np.bincount(xs, weights=t, minlength=8).astype(np.int64)   # tiles per eighth
            wpx = int(cnt.max())
            tot = int(t.sum())
            ws = np.repeat(perm, t)
            first = np.cumsum(t) - t                            # first work item of every sequence
            wt = np.arange(tot, dtype=np.int64) - np.repeat(first, t)
            seg0 = np.cumsum(cnt) - cnt                         # first work item of every eighth
            wx = np.repeat(xs, t)
            dest = wx * wpx + (np.arange(tot, dtype=np.int64) - seg0[wx])
            work = np.empty((8 * wpx, 2), np.int32)
            work[:, 0] = -1
            work[:, 1] = 0
            work[dest, 0] = ws
            work[dest, 1] = wt
        else:
            work = np.zeros((0, 2), np.int32)
        self.num_work = int(work.shape[0])
        # token row of the last position (CLS / last node) of every sequence: the pooled row
        last_row = desc[:, 0].astype(np.int64) + (desc[:, 1].astype(np.int64) - 1) * self.row_stride
        if torch.device(gs.device).type == "cuda":
            # ONE pinned staging buffer from a ring (pin_memory() per array cost ~30 us each) and ONE non_blocking H2D copy: a pageable
            # copy would block the host until everything already queued on the stream has drained (a 10 ms/step stall once)
            nd, nl, nw = desc.size * 4, last_row.size * 8, work.size * 4
            o_l = (nd + 15) // 16 * 16
            o_w = (o_l + nl + 15) // 16 * 16
            nbytes = max(o_w + nw, 16)
            pinned, event = _staging(nbytes, gs.device)
            if pinned is None:   # (larger than a ring slot: its own pinned buffer)
                pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
            hb = pinned.numpy()
            hb[:nd].view(np.int32)[:] = desc.reshape(-1)
            hb[o_l:o_l + nl].view(np.int64)[:] = last_row
            if nw:
                hb[o_w:o_w + nw].view(np.int32)[:] = work.reshape(-1)
            db = torch.empty(nbytes, dtype=torch.uint8, device=gs.device)
            db.copy_(pinned[:nbytes], non_blocking=True)
            if event is not None:
                event.record()
            self._dev = db
            self.desc = db[:nd].view(torch.int32).view(B, 4)
            self.last_rows = db[o_l:o_l + nl].view(torch.int64)
            self.work = db[o_w:o_w + nw].view(torch.int32).view(-1, 2) if self.num_work else None
        else:
            self.work = torch.from_numpy(work)
            self.desc = torch.from_numpy(desc)
            self.last_rows = torch.from_numpy(last_row)
