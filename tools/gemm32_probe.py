#!/usr/bin/env python
"""fp32 wide-tile GEMM forward: time vs number of 64-row blocks (co-residency / rounds probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphtrans_amd import _lib as L
from graphtrans_amd.ops import _ptr, _stream
DEV = "cuda:0"
lib = L.lib()
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
N = K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for blocks in (64, 128, 256, 384, 512, 640, 768, 1024, 1536, 2048):
    M = blocks * 64
    x, w, b, y = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV), torch.randn(N, device=DEV), torch.empty(M, N, device=DEV)
    t = timeit(lambda: lib.gt_linear_fwd(L.GT_F32, L.GT_F32, L.GT_F32, _ptr(x), _ptr(w), _ptr(b), _ptr(y), M, N, K, 0, 0.0, 0, _stream()))
    print(f"blocks {blocks:5d} M {M:7d}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF")
