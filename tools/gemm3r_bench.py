#!/usr/bin/env python
"""k_lin3r (rows straight into MFMA fragments, csrc/linear3r.h) against k_lin3 (both operands through the LDS): forward and un-gated dX
on the shapes of the benchmarked configurations.  usage: python tools/gemm3r_bench.py   (GPU box; A/B against another build: GT_LIB_PATH)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from graphtrans_amd.w3 import W3Images
from test_hip_linear3x import bwd_all, dx_of, fwd

DEV = "cuda:0"


def timeit(fn, n=100):
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


def main():
    tag = "k_lin3r"
    for M, N, K in [(31598, 300, 300), (131072, 256, 256), (16000, 272, 272), (31598, 600, 300), (31598, 300, 600), (12800, 300, 300)]:
        x = torch.randn(M, K, device=DEV)
        W = torch.randn(N, K, device=DEV) / K ** 0.5
        b = torch.randn(N, device=DEV)
        dy = torch.randn(M, N, device=DEV)
        a1 = torch.randn(M, K, device=DEV)
        imgs = W3Images([W])
        imgs.build()
        fl = 2.0 * M * N * K
        t = timeit(lambda: fwd(x, W, b, imgs))
        d = timeit(lambda: dx_of(x, W, dy, None, None, None, imgs))
        da = timeit(lambda: dx_of(x, W, dy, None, a1, None, imgs))
        dwt = timeit(lambda: bwd_all(x, W, dy, None, imgs), n=50)   # dX + dW + db + reduce
        print(f"{tag} {M:7d} x {N:4d} x {K:4d}: fwd {t:6.1f} us ({fl / t / 1e6:6.1f} TF, {fl / t / 1e6 / 416.7:.3f} of bf16x6)  dX {d:6.1f} us  dX+addend {da:6.1f} us  dX+dW+db+reduce {dwt:6.1f} us (dW part ~{dwt - d:5.1f}, {fl / max(dwt - d, 1e-3) / 1e6 / 416.7:.3f})", flush=True)


if __name__ == "__main__":
    main()
