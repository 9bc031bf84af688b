#!/usr/bin/env python
"""Micro-benchmarks of individual C-ABI ops on the GPU box (HIP-event timing, L2-warm, 30 iters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphtrans_amd import ops

DEV = "cuda:0"


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def bench_linear():
    shapes = [(256, 600, 300, "f32s"), (256, 300, 600, "f32s"), (6651, 600, 300, "f32s"), (31598, 300, 300, "f32s"), (31598, 128, 600, "f32s"), (31855, 384, 128, "bf16"), (31855, 128, 128, "bf16"),
              (31855, 512, 128, "bf16"), (31855, 128, 512, "bf16")]
    print(f"{'shape':28s} {'mode':6s} {'fwd us':>8s} {'dx us':>8s} {'dw us':>8s} {'torch fwd':>10s} {'torch bwd':>10s}  fwd TF/s  min-bytes us")
    for M, N, K, mode in shapes:
        for compute in (torch.bfloat16, torch.float32):
            if mode == "bf16" and compute == torch.float32:
                continue
            ops.set_matmul_dtype(compute)
            sdt = torch.bfloat16 if mode == "bf16" else torch.float32
            x = torch.randn(M, K, device=DEV).to(sdt).requires_grad_(True)
            w = torch.randn(N, K, device=DEV).requires_grad_(True)
            b = torch.randn(N, device=DEV).requires_grad_(True)
            g = torch.randn(M, N, device=DEV).to(sdt)
            L = __import__("graphtrans_amd._lib", fromlist=["x"])
            from graphtrans_amd.ops import _ptr, _stream, _dtype_code
            cc = L.GT_BF16 if compute == torch.bfloat16 else L.GT_F32
            y = torch.empty(M, N, device=DEV, dtype=sdt)
            dx = torch.empty_like(x); dw = torch.empty(N, K, device=DEV); db = torch.empty(N, device=DEV)
            lib = L.lib()
            wsb = lib.gt_linear_bwd_workspace_bytes(cc, M, N, K)
            ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
            xd, wd, bd = x.detach(), w.detach(), b.detach()
            f_fwd = lambda: lib.gt_linear_fwd(_dtype_code(xd), _dtype_code(y), cc, _ptr(xd), _ptr(wd), _ptr(bd), _ptr(y), M, N, K, 0, 0.0, 0, _stream())
            f_dx = lambda: lib.gt_linear_bwd(_dtype_code(xd), _dtype_code(g), cc, _ptr(xd), _ptr(wd), _ptr(g), None, None, None, _ptr(dx), None, None, M, N, K, 0.0, _ptr(ws), wsb, _stream())
            f_dw = lambda: lib.gt_linear_bwd(_dtype_code(xd), _dtype_code(g), cc, _ptr(xd), _ptr(wd), _ptr(g), None, None, None, None, _ptr(dw), _ptr(db), M, N, K, 0.0, _ptr(ws), wsb, _stream())
            t_f, t_dx, t_dw = timeit(f_fwd), timeit(f_dx), timeit(f_dw)
            wt = wd.to(sdt); bt = bd.to(sdt)
            t_tf = timeit(lambda: torch.nn.functional.linear(xd, wt, bt))
            def tb():
                gx = g @ wt
                gw = g.t() @ xd
                return gx, gw
            t_tb = timeit(tb)
            elt = 2 if sdt == torch.bfloat16 else 4
            minb = (M * K * elt + M * N * elt) / 6.0e12 * 1e6
            print(f"{str((M,N,K)):28s} {mode}/{'b' if compute==torch.bfloat16 else 'f'} {t_f:8.1f} {t_dx:8.1f} {t_dw:8.1f} {t_tf:10.1f} {t_tb:10.1f}  {2*M*N*K/t_f/1e6:8.1f}  {minb:8.1f}")
    ops.set_matmul_dtype(torch.float32)


if __name__ == "__main__":
    bench_linear()
