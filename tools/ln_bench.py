"""Micro-benchmark: LayerNorm backward (gt_layernorm_bwd: d(LN(resid + dropout(x)))) at the encoder's token-row shapes, bf16 and fp32 rows.
python tools/ln_bench.py [rows] [dim]"""
import sys
import torch
sys.path.insert(0, ".")
from graphtrans_amd import _lib
from graphtrans_amd.graph import _stream

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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 31598
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    lib = _lib.lib()
    for name, dt, code in (("bf16", torch.bfloat16, 1), ("fp32", torch.float32, 0)):
        for p in (0.0, 0.3):
            x = torch.randn(M, D, device=DEV).to(dt)
            r = torch.randn(M, D, device=DEV).to(dt)
            dy = torch.randn(M, D, device=DEV).to(dt)
            w = torch.ones(D, device=DEV)
            y = torch.empty_like(x)
            mu, rs = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
            _lib.launch("gt_layernorm_fwd", code, _p(x), _p(r), _p(w), _p(torch.zeros(D, device=DEV)), 1e-5, p, 7, M, D, _p(y), _p(mu), _p(rs), _stream())
            dx, dr = torch.empty_like(x), torch.empty_like(x)
            dw, db = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
            nb = int(lib.gt_layernorm_bwd_workspace_bytes(M, D))
            ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
            st = _stream()
            f = lambda: _lib.launch("gt_layernorm_bwd", code, _p(x), _p(r), _p(dy), _p(w), _p(mu), _p(rs), p, 7, M, D, _p(dx), _p(dr), _p(dw), _p(db), _p(ws), nb, st)
            t = timeit(f)
            mb = 5 * M * D * x.element_size() / 1e6
            print(f"LN bwd {name} rows {M} dim {D} dropout {p}: {t:6.1f} us (incl. the finish launch)  {mb / t / 1e6 * 1e6 / 1e3:.2f} TB/s on {mb:.0f} MB")


if __name__ == "__main__":
    main()
