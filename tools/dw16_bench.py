"""Micro-benchmark: bf16 weight gradient dW = dY^T X (+ db) at the encoder shapes (csrc/linear_dw16.h against the tiled k_linear_dw;
A/B against another build: GT_LIB_PATH).  python tools/dw16_bench.py [M]"""
import sys
import torch
sys.path.insert(0, ".")
from graphtrans_amd import _lib
from graphtrans_amd.graph import _stream

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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32500
    shapes = [(384, 128), (128, 128), (512, 128), (128, 512)] if M < 100000 else [(768, 256), (256, 256), (1024, 256), (256, 1024)]
    lib = _lib.lib()
    tot = 0.0
    for N, K in shapes:
        x = torch.randn(M, K, device=DEV).to(BF)
        W = torch.randn(N, K, device=DEV) / K ** 0.5
        dy = torch.randn(M, N, device=DEV).to(BF)
        dw = torch.empty(N, K, device=DEV)
        db = torch.empty(N, device=DEV)
        ws_bytes = lib.gt_linear_bwd_workspace_bytes(GT_BF16, M, N, K)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        st = _stream()
        d = lambda: _lib.launch("gt_linear_bwd_ld2", GT_BF16, GT_BF16, GT_BF16, _p(x), _p(W), _p(dy), None, None, None, None, _p(dw), _p(db),
                                M, N, K, K, N, 0.0, _p(ws), ws_bytes, st)
        t = timeit(d)
        tot += t
        ref = dy.double().t() @ x.double()
        err = ((dw.double() - ref).norm() / ref.norm()).item()
        mb = (M * N * 2 + M * K * 2 + N * K * 4) / 1e6
        print(f"M={M} N={N} K={K}: dW+db+reduce {t:7.1f} us  ({mb / t / 1e6 * 1e6 / 1e3:.2f} TB/s algorithmic, rel err {err:.1e})")
    print(f"sum over the four shapes of an encoder layer: {tot:.1f} us")


if __name__ == "__main__":
    main()
