"""Stand-alone timings of the Code2-PNA tower GEMMs (grouped launches, T = 4, F = 68, 15 945 nodes): forward, dX, dW of the pre stack
(K = 68 -> 136) and the post stack (K = 340 -> 204), exact-fp32 tiled kernels against the bf16x6 ones on bound images.
usage (GPU box): python tools/pna_gemm_bench.py"""
import torch
from graphtrans_amd import _lib, w3
from graphtrans_amd.graph import _stream

DEV, GT_F32 = "cuda:0", 0


def _p(t):
    return None if t is None else t.data_ptr()


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    M, T = 15945, 4
    for name, K, N in (("pre", 68, 136), ("post", 340, 204)):
        x = torch.randn(M, T * K, device=DEV)
        W = (torch.randn(T, N, K, device=DEV) / K ** 0.5).contiguous()
        b = torch.randn(T, N, device=DEV)
        dy = torch.randn(M, T * N, device=DEV)
        y, dx = torch.empty(M, T * N, device=DEV), torch.empty(M, T * K, device=DEV)
        dw, db = torch.empty(T, N, K, device=DEV), torch.empty(T, N, device=DEV)
        imgs = w3.W3Images([W[t] for t in range(T)])
        imgs.build()
        ws_bytes = _lib.lib().gt_linear_bwd_grouped_workspace_bytes(GT_F32, M, N, K, T)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=DEV)
        fwd = lambda: _lib.launch("gt_linear_fwd_grouped", GT_F32, GT_F32, GT_F32, _p(x), _p(W), _p(b), _p(y), M, N, K, T * K, T * N, T, K, N, 0, 0.0, 0, _stream())
        bdx = lambda: _lib.launch("gt_linear_bwd_grouped", GT_F32, GT_F32, GT_F32, _p(x), _p(W), _p(dy), None, None, None, _p(dx), None, None,
                                  M, N, K, T * K, T * N, T, K, N, 0.0, _p(ws), ws_bytes, _stream())
        bdw = lambda: _lib.launch("gt_linear_bwd_grouped", GT_F32, GT_F32, GT_F32, _p(x), _p(W), _p(dy), None, None, None, None, _p(dw), _p(db),
                                  M, N, K, T * K, T * N, T, K, N, 0.0, _p(ws), ws_bytes, _stream())
        flops = 2.0 * M * N * K * T
        for label, ctx in (("exact", None), ("bf16x6", imgs)):
            if ctx is not None:
                with ctx.bound():
                    t = [timed(fwd), timed(bdx), timed(bdw)]
            else:
                t = [timed(fwd), timed(bdx), timed(bdw)]
            print(f"{name:5s} {label:7s} fwd {t[0]:7.1f} us ({flops / t[0] / 1e6:6.1f} TF)  dX {t[1]:7.1f} us ({flops / t[1] / 1e6:6.1f} TF)  "
                  f"dW+reduce {t[2]:7.1f} us ({flops / t[2] / 1e6:6.1f} TF)", flush=True)


if __name__ == "__main__":
    main()
