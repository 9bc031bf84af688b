#!/usr/bin/env python
"""Two unrelated entry points on two streams at once (weight-gradient GEMM on the side stream, LayerNorm backward on the
main stream, no shared buffers): every output must be bitwise reproducible."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphtrans_amd import _lib
from graphtrans_amd._lib import GT_BF16
lib = _lib.lib()
dev = "cuda:0"
R, d = 32548, 128
torch.manual_seed(0)
x = torch.randn(R, d, device=dev).bfloat16(); dy = torch.randn(R, 3 * d, device=dev).bfloat16(); w = torch.randn(3 * d, d, device=dev)
dx = torch.zeros(R, d, device=dev).bfloat16(); dw = torch.empty(3 * d, d, device=dev); db = torch.empty(3 * d, device=dev)
wsb = lib.gt_linear_bwd_workspace_bytes(GT_BF16, R, 3 * d, d); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
lx = torch.randn(R, d, device=dev).bfloat16(); lr = torch.randn(R, d, device=dev).bfloat16(); ldy = torch.randn(R, d, device=dev).bfloat16()
lw = torch.randn(d, device=dev); mean = torch.zeros(R, device=dev); rstd = torch.ones(R, device=dev)
o1 = torch.empty_like(lx); o2 = torch.empty_like(lx); gw = torch.empty(d, device=dev); gb = torch.empty(d, device=dev)
lnb = lib.gt_layernorm_bwd_workspace_bytes(R, d); lws = torch.empty(lnb, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(); st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
first, bad = None, {}
for it in range(300):
    dx.zero_(); o1.zero_(); o2.zero_()
    _lib.check(lib.gt_overlap_dw_begin(st, side.cuda_stream), "b")
    _lib.check(lib.gt_linear_bwd(GT_BF16, GT_BF16, GT_BF16, p(x), p(w), p(dy), None, p(dx), None, p(dx), p(dw), p(db), R, 3 * d, d, 0.0, p(ws), wsb, st), "lin")
    _lib.check(lib.gt_layernorm_bwd(GT_BF16, p(lx), p(lr), p(dx), p(lw), p(mean), p(rstd), 0.0, 0, R, d, p(o1), p(o2), p(gw), p(gb), p(lws), lnb, st), "ln")
    _lib.check(lib.gt_overlap_dw_end(), "e")
    torch.cuda.synchronize()
    cur = dict(dw=dw.clone(), db=db.clone(), gw=gw.clone(), gb=gb.clone(), o1=o1.clone(), o2=o2.clone(), dx=dx.clone())
    if first is None:
        first = cur
    else:
        for k in cur:
            if not torch.equal(cur[k], first[k]):
                bad[k] = bad.get(k, 0) + 1
print("mismatching outputs over 299 repeats:", bad or "none")
