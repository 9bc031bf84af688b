import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


class Golden:
    """One tests/golden/*.npz fixture (generated from the reference by oracle/make_golden.py)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.name = name
        self.meta = json.loads(str(z["meta"]))
        self.inputs, self.sd, self.outs, self.gin, self.gsd = {}, {}, {}, {}, {}
        for k in z.files:
            if k == "meta":
                continue
            pre, rest = k.split(".", 1)
            t = torch.from_numpy(z[k])
            {"in": self.inputs, "sd": self.sd, "out": self.outs, "gin": self.gin, "gsd": self.gsd}[pre][rest] = t
        self.out_list = [self.outs[str(i)] for i in range(len(self.outs))] if all(k.isdigit() for k in self.outs) else None

    def batch(self):
        from graphtrans_amd.data import Batch

        d = {k: self.inputs[k] for k in ("x", "edge_index", "edge_attr", "batch", "node_depth") if k in self.inputs}
        d.setdefault("edge_attr", None)
        b = Batch(**d)
        adj = [self.inputs[k].numpy() for k in sorted((k for k in self.inputs if k.startswith("adj")), key=lambda s: int(s[3:]))]
        if adj:
            b.adj_list = adj
        return b

    def args(self):
        from types import SimpleNamespace

        return SimpleNamespace(**self.meta["args"])


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def assert_close(a, b, atol=1e-4, rtol=1e-4, what=""):
    """|a-b| <= atol*max(1, max|b|) + rtol*|b| elementwise: the 1e-4 fp32 bar of BASELINE.json's
    north_star, taken relative to the tensor's scale (gradients of the randomised fixtures reach
    1e4, where fp32 summation-order noise alone is ~1e-3 absolute)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    tol = atol * scale + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: max abs err {err.max().item():.3e} (tol {atol}+{rtol}*|ref|), {int(bad.sum())}/{bad.numel()} bad"
