#!/usr/bin/env python
"""BatchNorm forward (+ ReLU + the next layer's virtual-node add) and backward: the one-launch scheme (csrc/norm_coop.h) against the
three-launch scheme, alone on the chip.  usage: python tools/bn_bench.py   (GPU box)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from graphtrans_amd import _lib

DEV = "cuda:0"
L = _lib.lib()


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream(0)
    s.record(cur)
    for _ in range(n):
        fn()
    e.record(cur)
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


for rows, D, B in [(31598, 300, 256), (6611, 300, 256), (6611, 600, 256), (15800, 272, 128), (4000, 300, 32)]:
    x = torch.randn(rows, D, device=DEV)
    g = torch.randn(rows, D, device=DEV)
    w, b = torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV)
    vn = torch.randn(B, D, device=DEV)
    idx = torch.sort(torch.randint(0, B, (rows,), dtype=torch.int32)).values.to(DEV)
    rm, rv = torch.zeros(D, device=DEV), torch.ones(D, device=DEV)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    mean, rstd, dw, db = (torch.empty(D, device=DEV) for _ in range(4))
    wsb = L.gt_batchnorm_workspace_bytes(rows, D)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    slots = torch.zeros(4096 * 16, dtype=torch.int32, device=DEV)   # zeroed barrier slots: enough for every call of a timing loop
    ptr = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fwd = lambda: _lib.launch("gt_batchnorm_fwd_bcast", 0, ptr(x), ptr(w), ptr(b), ptr(rm), ptr(rv), None, 0.1, 1e-5, 1, 1, None, ptr(vn), ptr(idx), None,
                              rows, D, ptr(y), ptr(mean), ptr(rstd), 0.0, 0, ptr(ws), wsb, st)
    bwd = lambda: _lib.launch("gt_batchnorm_bwd", 0, ptr(x), ptr(g), ptr(w), ptr(b), ptr(mean), ptr(rstd), 1, 1, rows, D, ptr(dx), ptr(dw), ptr(db),
                              0.0, 0, ptr(ws), wsb, st)
    out = []
    for on in (1, 0):
        L.gt_bn_coop_set(on)
        res = []
        for fn in (fwd, bwd):
            if on:   # slots from a pool (as the whole-model driver does): no memset in front of the launches
                slots.zero_()
                L.gt_bn_coop_slots(ptr(slots), 4096)
            res.append(timeit(fn))
            L.gt_bn_coop_slots(None, 0)
        out.append(res)
    L.gt_bn_coop_set(-1)
    mb = rows * D * 4 / 1e6
    print(f"{rows:6d} x {D:4d}: forward one launch {out[0][0]:6.1f} us ({2 * mb / out[0][0] * 1e-3:5.2f} TB/s on 2 x {mb:.0f} MB)  three {out[1][0]:6.1f} us | "
          f"backward one launch {out[0][1]:6.1f} us  three {out[1][1]:6.1f} us", flush=True)
