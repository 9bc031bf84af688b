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
        self.outs64, self.gin64, self.gsd64 = {}, {}, {}   # float64 twin run of the same reference code (G12)
        for k in z.files:
            if k == "meta":
                continue
            pre, rest = k.split(".", 1)
            t = torch.from_numpy(z[k])
            {"in": self.inputs, "sd": self.sd, "out": self.outs, "gin": self.gin, "gsd": self.gsd,
             "out64": self.outs64, "gin64": self.gin64, "gsd64": self.gsd64}[pre][rest] = t
        self.out_list = [self.outs[str(i)] for i in range(len(self.outs))] if all(k.isdigit() for k in self.outs) else None
        self.out64_list = [self.outs64[str(i)] for i in range(len(self.outs64))]

    def ref_noise(self, kind, key):
        """max|fp32 reference - float64 reference| / max(1, max|float64|) of one tensor: the rounding noise the
        reference's OWN fp32 evaluation carries (kind in out / gsd / gin)."""
        a = {"out": self.outs, "gsd": self.gsd, "gin": self.gin}[kind][key].double()
        b = {"out": self.outs64, "gsd": self.gsd64, "gin": self.gin64}[kind][key]
        return float((a - b).abs().max()) / max(1.0, float(b.abs().max()))

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


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm()) / max(float(b.norm()), 1e-300)


def quantile_err(a, b, q=0.98):
    """q-quantile of |a - b| relative to max|b|: an elementwise error measure that ignores the (1 - q) worst elements.
    Used for gradients at REAL model sizes, where single ReLU gate flips (a pre-activation within fp32 rounding of
    zero, among millions of units) move one row / one element of a gradient by O(1e-2) in ANY fp32 evaluation --
    the reference's own included (tools/flip_count.py, tests/test_hip_configs.py)."""
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    e = (a - b).abs() / max(float(b.abs().max()), 1e-300)
    if e.numel() < 50:
        return float(e.max())
    k = min(e.numel(), max(1, int(round(q * e.numel()))))
    return float(e.kthvalue(k).values)


@pytest.fixture(autouse=True)
def _exercise_dw_overlap_at_test_sizes(request):
    """The fused backward forks its weight-gradient GEMMs onto a third stream only for batches above
    engine.DW_OVERLAP_MIN_ELEMS (small steps are host-bound); the GPU tests run tiny batches and must cover that path."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from graphtrans_amd import engine
    old = engine.DW_OVERLAP_MIN_ELEMS
    engine.DW_OVERLAP_MIN_ELEMS = 0
    yield
    engine.DW_OVERLAP_MIN_ELEMS = old
