#!/usr/bin/env python
"""bf16 GEMM micro-benchmark (the encoder layers' linears, bf16 rows): python tools/gemm16_bench.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphtrans_amd import _lib as L
from graphtrans_amd.ops import _ptr, _stream

DEV = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
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


B = L.GT_BF16
for M, N, K in ((32500, 384, 128), (32500, 128, 128), (32500, 512, 128), (32500, 128, 512), (131000, 768, 256), (131000, 256, 1024)):
    x, w, b = torch.randn(M, K, device=DEV).bfloat16(), torch.randn(N, K, device=DEV), torch.randn(N, device=DEV)
    g, y, dx = torch.randn(M, N, device=DEV).bfloat16(), torch.empty(M, N, device=DEV, dtype=torch.bfloat16), torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
    dw, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    wsb = lib.gt_linear_bwd_workspace_bytes(B, M, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    f = lambda: lib.gt_linear_fwd(B, B, B, _ptr(x), _ptr(w), _ptr(b), _ptr(y), M, N, K, 1, 0.0, 0, _stream())
    fdx = lambda: lib.gt_linear_bwd(B, B, B, _ptr(x), _ptr(w), _ptr(g), _ptr(y), None, None, _ptr(dx), None, None, M, N, K, 0.0, _ptr(ws), wsb, _stream())
    fdw = lambda: lib.gt_linear_bwd(B, B, B, _ptr(x), _ptr(w), _ptr(g), _ptr(y), None, None, None, _ptr(dw), _ptr(db), M, N, K, 0.0, _ptr(ws), wsb, _stream())
    t = [timeit(f), timeit(fdx), timeit(fdw)]
    fl = 2.0 * M * N * K
    by = 2.0 * M * (N + K)
    print(f"{(M, N, K)}: fwd {t[0]:.1f} us ({fl / t[0] / 1e6:.0f} TF, {by / t[0] / 1e3:.0f} GB/s)  dx {t[1]:.1f} us ({fl / t[1] / 1e6:.0f} TF)  dw {t[2]:.1f} us ({fl / t[2] / 1e6:.0f} TF)")
