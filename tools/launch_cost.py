#!/usr/bin/env python
"""Host cost of one C-ABI call + kernel launch (ctypes path), measured on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphtrans_amd import _lib
from graphtrans_amd.graph import _stream
L = _lib.lib()
a = torch.zeros(1024, device="cuda:0"); b = torch.zeros(1024, device="cuda:0")
st = _stream()
for n in (2000,):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        L.gt_copy2d(a.data_ptr(), 64, b.data_ptr(), 64, 64, 16, st)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"ctypes call + launch: {1e6*(t1-t0)/n:.2f} us/call host, {1e6*(t2-t0)/n:.2f} us/call incl. drain")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        L.gt_version()
    t1 = time.perf_counter()
    print(f"ctypes call only: {1e6*(t1-t0)/n:.2f} us/call")
    x = torch.zeros(1024, device="cuda:0")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        x.add_(1.0)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"torch add_: {1e6*(t1-t0)/n:.2f} us/call host")
