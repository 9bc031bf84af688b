"""usage (GPU box): python tools/bn_mid_bench.py   -- BatchNorm forward / backward at mid row counts, one-launch kernels against the
three-launch scheme (GT_BN_MID_ROWS=0 in a second process)"""
import ctypes as C, os, subprocess, sys, torch
def run():
    from graphtrans_amd import _lib
    dev = "cuda:0"
    ptr = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rows, D in [(6611, 300), (6611, 600), (4000, 300), (8000, 300), (2000, 128)]:
        x = torch.randn(rows, D, device=dev); w = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
        rm, rv = torch.zeros(D, device=dev), torch.ones(D, device=dev)
        y = torch.empty_like(x); g = torch.randn_like(x); dx = torch.empty_like(x)
        mean, rstd, dw, db = (torch.empty(D, device=dev) for _ in range(4))
        wsb = _lib.lib().gt_batchnorm_workspace_bytes(rows, D); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        def fwd(): _lib.launch("gt_batchnorm_fwd_bcast", 0, ptr(x), ptr(w), ptr(b), ptr(rm), ptr(rv), None, 0.1, 1e-5, 1, 1, None, None, None, None, rows, D, ptr(y), ptr(mean), ptr(rstd), 0.0, 0, ptr(ws), wsb, st)
        def bwd(): _lib.launch("gt_batchnorm_bwd", 0, ptr(x), ptr(g), ptr(w), ptr(b), ptr(mean), ptr(rstd), 1, 1, rows, D, ptr(dx), ptr(dw), ptr(db), 0.0, 0, ptr(ws), wsb, st)
        for name, f in (("fwd", fwd), ("bwd", bwd)):
            for _ in range(10): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): f()
            e1.record(); torch.cuda.synchronize()
            print(f"{os.environ.get('GT_BN_MID_ROWS', 'mid')} {rows}x{D} {name}: {e0.elapsed_time(e1) * 5:.2f} us")
if __name__ == "__main__":
    if len(sys.argv) > 1: run()
    else:
        for env in ({}, {"GT_BN_MID_ROWS": "0"}):
            subprocess.run([sys.executable, __file__, "x"], env={**os.environ, **env})
