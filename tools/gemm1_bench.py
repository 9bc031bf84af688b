"""Micro-benchmark: weight-stationary bf16 encoder GEMMs (csrc/linear1.h, bound images) against the tiled kernels (unbound), forward and dX,
at the encoder shapes of Code2 (M = 32 k tokens) and ER (M = 131 k).  python tools/gemm1_bench.py [M]"""
import sys
import torch
sys.path.insert(0, ".")
from graphtrans_amd import _lib
from graphtrans_amd.graph import _stream
from graphtrans_amd.w3 import W1Images

GT_BF16 = 1
BF = torch.bfloat16
DEV = "cuda:0"


def _p(t):
    return None if t is None else t.data_ptr()


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
    shapes = [(384, 128), (128, 128), (512, 128), (128, 512)] if M < 100000 else [(768, 256), (256, 256), (1024, 256), (256, 1024)]
    lib = _lib.lib()
    for N, K in shapes:
        x = torch.randn(M, K, device=DEV).to(BF)
        W = torch.randn(N, K, device=DEV) / K ** 0.5
        b = torch.randn(N, device=DEV)
        y = torch.empty(M, N, dtype=BF, device=DEV)
        dy = torch.randn(M, N, device=DEV).to(BF)
        dx = torch.empty(M, K, dtype=BF, device=DEV)
        add = torch.randn(M, K, device=DEV).to(BF)
        ws_bytes = lib.gt_linear_bwd_workspace_bytes(GT_BF16, M, N, K)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        imgs = W1Images([W])
        imgs.build()
        st = _stream()
        f = lambda: _lib.launch("gt_linear_fwd_ld2", GT_BF16, GT_BF16, GT_BF16, _p(x), _p(W), _p(b), _p(y), M, N, K, K, N, 1, 0.3, 7, st)
        d = lambda: _lib.launch("gt_linear_bwd_ld2", GT_BF16, GT_BF16, GT_BF16, None, _p(W), _p(dy), None, _p(add), None, _p(dx), None, None,
                                M, N, K, K, N, 0.0, _p(ws), ws_bytes, st)
        t_f0, t_d0 = timeit(f), timeit(d)
        with imgs.bound():
            prev = _lib.option_set("lin_ring", 1)      # k_lin1 (the tiled kernel where it does not cover the shape)
            t_f1, t_d1 = timeit(f), timeit(d)
            _lib.option_set("lin_ring", 2)             # k_lin2: both operands through the LDS ring
            t_f2, t_d2 = timeit(f), timeit(d)
            _lib.option_set("lin_ring", prev)
        gb_f = (M * K * 2 + M * N * 2) / 1e3
        gb_d = (M * N * 2 + 2 * M * K * 2) / 1e3
        print(f"M={M} N={N} K={K}: fwd tiled {t_f0:7.1f} us  stationary {t_f1:7.1f} us  ring {t_f2:7.1f} us ({gb_f / t_f2 / 1e3:.2f} TB/s) | "
              f"dx tiled {t_d0:7.1f} us  stationary {t_d1:7.1f} us  ring {t_d2:7.1f} us ({gb_d / t_d2 / 1e3:.2f} TB/s)")
    t = timeit(lambda: imgs.build())
    print(f"image build (1 weight, both directions): {t:.1f} us")


if __name__ == "__main__":
    main()
