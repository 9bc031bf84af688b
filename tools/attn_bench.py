#!/usr/bin/env python
"""Attention kernel micro-benchmark: where does the time go -- the serial key loop of the longest sequence, or throughput?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from graphtrans_amd import ops, synth
from test_hip_attention import make_layout

DEV = "cuda:0"


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def case(name, lens, d=128, nhead=4, dtype=torch.bfloat16, lpt=None, p=0.0):
    if lpt is None:
        case(name + " [seq order]", lens, d, nhead, dtype, False, p)
        case(name + " [longest first]", lens, d, nhead, dtype, True, p)
        return
    lay = make_layout("packed", lens)
    tiles = (np.asarray(lens) + 63) // 64
    if lpt:   # the work list graph.SeqLayout builds
        class G: pass
        g = G(); g.sizes = np.asarray(lens); g.B = len(lens); g.device = "cpu"
        from graphtrans_amd.graph import SeqLayout
        work = SeqLayout(g, "packed", 10 ** 9, False).work.numpy()
    else:
        work = np.stack([np.repeat(np.arange(len(lens)), tiles), np.concatenate([np.arange(t) for t in tiles])], 1).astype(np.int32)
    lay.work = torch.from_numpy(work).to(DEV)
    lay.num_work = int(work.shape[0])
    qkv = torch.randn(lay.rows, 3 * d, device=DEV).to(dtype).requires_grad_(True)
    w = torch.randn(lay.rows, d, device=DEV).to(dtype)
    kw = dict(dropout_p=p, seed=7) if p else {}
    out = ops.attention(qkv, lay, nhead, **kw)
    t_f = timeit(lambda: ops.attention(qkv.detach(), lay, nhead, **kw))
    def fb():
        o = ops.attention(qkv, lay, nhead, **kw)
        o.backward(w)
    t_fb = timeit(fb)
    fl = 4.0 * float((np.asarray(lens, dtype=np.float64) ** 2).sum()) * d
    # kernel durations from the library's own HIP-event brackets (the host wrapper costs ~15 us per call)
    from graphtrans_amd import _lib
    _lib.profile_enable(2)
    for _ in range(10):
        fb()
    rec = _lib.profile_records()
    _lib.profile_enable(0)
    agg = {}
    for nm, ms, _ in rec:
        agg.setdefault(nm, []).append(ms * 1e3)
    kt = "  ".join(f"{k} {np.median(v):.1f} us" for k, v in agg.items())
    print(f"{name:34s} kernels: {kt}")
    print(f"{name:34s} seqs {len(lens):4d} rows {lay.rows:6d} max {max(lens):5d}: fwd {t_f:7.1f} us ({fl / t_f / 1e6:6.1f} TF)  fwd+bwd {t_fb:7.1f} us")


if __name__ == "__main__":
    case("one sequence of 1001", [1001])
    case("one sequence of 513", [513])
    case("one sequence of 126", [126])
    case("256 x 126", [126] * 256)
    case("1024 x 126", [126] * 1024)
    case("256 x 126 + one of 1001", [126] * 255 + [1001])
    b = synth.code2_like(B=256, seed=1000)
    n = np.minimum(torch.bincount(b.batch).numpy(), 1000) + 1
    case("Code2-like batch (seed 1000)", list(n))
    case("Code2-like batch, dropout 0.3", list(n), p=0.3)
    case("256 x 513 (ER), d256 h4", [513] * 256, d=256, nhead=4)
