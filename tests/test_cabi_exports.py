"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/graphtrans_hip.h declares (no compute calls: there is no GPU here); host
logic (sequence layouts, synthetic generators) matches the integer oracle."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO


def _declared_symbols():
    src = open(os.path.join(REPO, "include", "graphtrans_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gt_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from graphtrans_amd import _lib, build

    build.build()
    assert os.path.exists(_lib.LIB_PATH)
    h = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/graphtrans_hip.h but not exported"
    # the ctypes table binds exactly the declared entry points
    assert sorted(_lib.SIGNATURES) == names
    L = _lib.lib()
    assert L.gt_version() >= 100
    assert L.gt_graph_prep_workspace_bytes(10, 20, 2) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from graphtrans_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU or eager fallback"):
        _lib.lib()


def test_cpu_tensors_are_rejected():
    from graphtrans_amd.graph import GraphStructure

    with pytest.raises(RuntimeError, match="GPU only"):
        GraphStructure.build(torch.zeros((2, 0), dtype=torch.int64), torch.zeros(3, dtype=torch.int64))


class _FakeGS:
    def __init__(self, sizes):
        self.sizes = np.asarray(sizes, np.int64)
        self.B = len(sizes)
        self.device = "cpu"


@pytest.mark.parametrize("sizes,max_len", [((5, 1, 9, 3), 1000), ((5, 1, 9, 3, 12), 7), ((4, 4, 4), 4)])
def test_seq_layout_matches_pad_index_oracle(sizes, max_len):
    from graphtrans_amd.graph import SeqLayout
    from oracle.graph_struct import pad_index

    ptr = np.concatenate([[0], np.cumsum(sizes)])
    S, kept, first = pad_index(ptr, max_len)
    for with_cls in (False, True):
        pad = SeqLayout(_FakeGS(sizes), "padded", max_len, with_cls)
        assert pad.S == S and np.array_equal(pad.kept, kept)
        d = pad.desc_cpu
        assert np.all(d[:, 1] == S + with_cls) and np.array_equal(d[:, 3], kept + with_cls)
        assert np.array_equal(d[:, 2], S + with_cls - d[:, 3]) and pad.rows == (S + with_cls) * len(sizes)
        pk = SeqLayout(_FakeGS(sizes), "packed", max_len, with_cls)
        d = pk.desc_cpu
        assert np.array_equal(d[:, 0], np.concatenate([[0], np.cumsum(kept + with_cls)])[:-1])
        assert pk.rows == int((kept + with_cls).sum()) and pk.max_npos == int((kept + with_cls).max())
        assert np.array_equal(pk.last_rows.numpy(), d[:, 0] + d[:, 1] - 1)


def test_synthetic_batches_are_well_formed():
    from graphtrans_amd import synth

    for b in (synth.code2_like(8, 0), synth.molpcba_like(8, 0), synth.nci1_like(8, 0), synth.er_stress(2, 0, n=64, feat_dim=8)):
        n = b.num_nodes
        assert b.edge_index.dtype == torch.int64 and b.edge_index.min() >= 0 and b.edge_index.max() < n
        assert torch.all(b.batch[1:] >= b.batch[:-1]) and int(b.batch[-1]) + 1 == b.num_graphs
        # edges never cross graphs (block-diagonal collation)
        assert torch.equal(b.batch[b.edge_index[0]], b.batch[b.edge_index[1]])
    a, c = synth.code2_like(4, 7), synth.code2_like(4, 7)
    assert torch.equal(a.edge_index, c.edge_index) and torch.equal(a.x, c.x)
