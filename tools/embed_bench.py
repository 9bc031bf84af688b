#!/usr/bin/env python
"""gt_embed_sum_bwd alone (C ABI, back-to-back launches, HIP events): cost of the table-gradient scatter per table mix."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from graphtrans_amd import _lib

N, D = 31598, 300
st = torch.cuda.current_stream().cuda_stream
lib = _lib.lib()


def run(name, rows_list, hot=0.0):
    T = len(rows_list)
    idx = [torch.randint(0, r, (N,)).cuda() for r in rows_list]
    if hot:
        for i in idx:
            i[torch.rand(N, device="cuda") < hot] = 1
    g = torch.randn(N, D).cuda()
    grads = [torch.empty(r, D).cuda() for r in rows_list]
    I64, P = C.c_int64 * T, C.c_void_p * T
    pidx, strides, clamp, rows_c = P(*[i.data_ptr() for i in idx]), I64(*[1] * T), I64(*[-1] * T), I64(*rows_list)
    dt = P(*[x.data_ptr() for x in grads])
    wsb = lib.gt_embed_sum_bwd_workspace_bytes(T, rows_c, D)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    f = lambda: _lib.check(lib.gt_embed_sum_bwd(T, pidx, strides, clamp, rows_c, g.data_ptr(), N, D, dt, ws.data_ptr(), wsb, st), "bwd")
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        f()
    e.record()
    torch.cuda.synchronize()
    t_fixed = s.elapsed_time(e) / 20 * 1e3
    # sorted path: gt_embed_sort (once per batch, beside the forward) + gt_embed_sum_bwd_sorted
    plan = torch.empty(lib.gt_embed_sort_plan_bytes(T, rows_c, N), dtype=torch.uint8, device="cuda")
    swb = lib.gt_embed_sort_workspace_bytes(T, rows_c, N)
    sws = torch.empty(swb, dtype=torch.uint8, device="cuda")
    bwb = lib.gt_embed_sum_bwd_sorted_workspace_bytes(T, N, D)
    bws = torch.empty(bwb, dtype=torch.uint8, device="cuda")
    fs = lambda: _lib.check(lib.gt_embed_sort(T, pidx, strides, clamp, rows_c, N, plan.data_ptr(), plan.numel(), sws.data_ptr(), swb, st), "sort")
    fb = lambda: _lib.check(lib.gt_embed_sum_bwd_sorted(T, rows_c, g.data_ptr(), N, D, plan.data_ptr(), dt, bws.data_ptr(), bwb, st), "bwd_sorted")
    out = []
    for fn in (fs, fb):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / 20 * 1e3)
    print(f"{name:44s} fixed-point atomics {t_fixed:7.1f} us | sort {out[0]:6.1f} us + sorted segments {out[1]:6.1f} us")


run("code2: type 98 + attr 10030 + depth 21", [98, 10030, 21])
run("attr 10030 only", [10030])
run("type 98 + depth 21", [98, 21])
run("type 98 only", [98])
run("depth 21 only", [21])
run("code2 tables, 80 % of nodes on one row each", [98, 10030, 21], hot=0.8)
