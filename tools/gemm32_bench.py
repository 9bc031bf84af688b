#!/usr/bin/env python
"""Exact-fp32 GEMM micro-benchmark (the message-passing linears): python tools/gemm32_bench.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphtrans_amd import _lib as L
from graphtrans_amd.ops import _ptr, _stream

DEV = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib = L.lib()


def timeit(fn, n=iters, warm=5):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, N, K in ((31598, 300, 300), (31598, 128, 600), (131072, 256, 256), (6651, 600, 300), (256, 25012, 128), (256, 600, 300)):
    x, w, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV), torch.randn(N, device=DEV)
    g, y, dx = torch.randn(M, N, device=DEV), torch.empty(M, N, device=DEV), torch.empty(M, K, device=DEV)
    dw, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    wsb = lib.gt_linear_bwd_workspace_bytes(L.GT_F32, M, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    f = lambda: lib.gt_linear_fwd(L.GT_F32, L.GT_F32, L.GT_F32, _ptr(x), _ptr(w), _ptr(b), _ptr(y), M, N, K, 0, 0.0, 0, _stream())
    fdx = lambda: lib.gt_linear_bwd(L.GT_F32, L.GT_F32, L.GT_F32, _ptr(x), _ptr(w), _ptr(g), None, None, None, _ptr(dx), None, None, M, N, K, 0.0, _ptr(ws), wsb, _stream())
    fdw = lambda: lib.gt_linear_bwd(L.GT_F32, L.GT_F32, L.GT_F32, _ptr(x), _ptr(w), _ptr(g), None, None, None, None, _ptr(dw), _ptr(db), M, N, K, 0.0, _ptr(ws), wsb, _stream())
    t = [timeit(f), timeit(fdx), timeit(fdw)]
    fl = 2.0 * M * N * K
    print(f"{(M, N, K)}: fwd {t[0]:.1f} us ({fl / t[0] / 1e6:.1f} TF)  dx {t[1]:.1f} us ({fl / t[1] / 1e6:.1f} TF)  dw {t[2]:.1f} us ({fl / t[2] / 1e6:.1f} TF)")
