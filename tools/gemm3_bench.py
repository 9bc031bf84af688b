#!/usr/bin/env python
"""k_lin3 (bf16x6) against k_lin32 (exact-fp32 MFMA): forward and dX on the shapes of the benchmarked configurations.
usage: python tools/gemm3_bench.py   (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from graphtrans_amd.w3 import W3Images
from test_hip_linear3x import dx_of, fwd

DEV = "cuda:0"


def timeit(fn, n=60):
    for _ in range(5):
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


for M, N, K in [(31598, 300, 300), (31598, 128, 600), (31598, 600, 128), (6700, 600, 300), (6700, 300, 600), (6700, 300, 300), (131072, 256, 256), (16000, 272, 272)]:
    x = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    dy = torch.randn(M, N, device=DEV)
    yf = torch.relu(torch.randn(M, N, device=DEV))
    imgs = W3Images([W])
    imgs.build()
    fl = 2.0 * M * N * K
    t3, t32 = timeit(lambda: fwd(x, W, b, imgs, act=1)), timeit(lambda: fwd(x, W, b, None, act=1))
    d3, d32 = timeit(lambda: dx_of(x, W, dy, yf, None, None, imgs)), timeit(lambda: dx_of(x, W, dy, yf, None, None, None))
    ti = timeit(lambda: imgs.build())
    print(f"{M:7d} x {N:4d} x {K:4d}: fwd bf16x6 {t3:6.1f} us ({fl / t3 / 1e6:6.1f} TF)  exact {t32:6.1f} us ({fl / t32 / 1e6:6.1f} TF) | "
          f"dX bf16x6 {d3:6.1f} us  exact(+transpose) {d32:6.1f} us | image build (fwd + T) {ti:5.1f} us", flush=True)
