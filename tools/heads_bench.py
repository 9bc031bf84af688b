"""usage (GPU box): PYTHONPATH=. python tools/heads_bench.py   -- the prediction heads (256 x 25 010 x 128) forward / dX / dW alone,
exact fp32 and bf16 operands (A/B against another build of the library: GT_LIB_PATH)"""
import ctypes as C, os, subprocess, sys, torch
def run():
    from graphtrans_amd import _lib
    dev = "cuda:0"
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M, N, K = 256, 25010, 128
    ld = (N + 3) // 4 * 4
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = torch.empty(M, ld, device=dev); dy = torch.randn(M, ld, device=dev); dx = torch.empty(M, K, device=dev)
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    for comp, cname in ((0, "fp32"), (1, "bf16")):
        wsb = _lib.lib().gt_linear_bwd_workspace_bytes(comp, M, N, K); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        def fwd(): _lib.launch("gt_linear_fwd_ld", 0, 0, comp, ptr(x), ptr(w), ptr(b), ptr(y), M, N, K, ld, 0, 0.0, 0, st)
        def bdx(): _lib.launch("gt_linear_bwd_ld", 0, 0, comp, ptr(x), ptr(w), ptr(dy), None, None, None, ptr(dx), None, None, M, N, K, ld, 0.0, ptr(ws), wsb, st)
        def bdw(): _lib.launch("gt_linear_bwd_ld", 0, 0, comp, ptr(x), ptr(w), ptr(dy), None, None, None, None, ptr(dw), ptr(db), M, N, K, ld, 0.0, ptr(ws), wsb, st)
        for name, f in (("fwd", fwd), ("dx", bdx), ("dw", bdw)):
            for _ in range(10): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): f()
            e1.record(); torch.cuda.synchronize()
            print(f"{cname} {name}: {e0.elapsed_time(e1) * 10:.2f} us")
        if comp == 0:
            ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
            print("  max |y - ref|", float((y[:, :N].double() - ref).abs().max()), " max |dx - ref|", float((dx.double() - dy[:, :N].double() @ w.double()).abs().max()))
if __name__ == "__main__":
    run()
