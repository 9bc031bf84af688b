"""Host half of the whole-model driver (csrc/model.hip), runnable without a GPU: the packed token layout built in C equals the numpy
statement of it (graphtrans_amd/graph.py:SeqLayout: the reference's pad_batch bookkeeping modules/utils.py:9-16 + the CLS position of
modules/transformer_encoder.py:50-55), and the ctypes mirrors of the driver's structs have the library's sizes."""
import ctypes as C

import numpy as np
import pytest


class _FakeGS:
    def __init__(self, sizes):
        self.sizes = np.asarray(sizes, np.int64)
        self.B = len(sizes)
        self.device = "cpu"


def _host_layout(sizes, max_len, cls):
    from graphtrans_amd import _lib
    L = _lib.lib()
    sizes = np.ascontiguousarray(sizes, np.int64)
    meta = (C.c_int64 * 6)()
    _lib.check(L.gt_seq_layout_packed_host(sizes.ctypes.data, sizes.size, max_len, cls, None, 0, meta), "size")
    buf = np.zeros(meta[5], np.uint8)
    _lib.check(L.gt_seq_layout_packed_host(sizes.ctypes.data, sizes.size, max_len, cls, buf.ctypes.data, buf.size, meta), "fill")
    B = sizes.size
    return (tuple(meta[:3]), buf[:B * 16].view(np.int32).reshape(B, 4), buf[meta[3]:meta[3] + B * 8].view(np.int64),
            buf[meta[4]:meta[4] + meta[2] * 8].view(np.int32).reshape(-1, 2))


@pytest.mark.parametrize("seed", range(6))
def test_host_layout_equals_numpy_layout(seed):
    from graphtrans_amd.graph import SeqLayout
    rng = np.random.default_rng(seed)
    for _ in range(12):
        B = int(rng.integers(1, 300))
        sizes = rng.integers(1, 700, size=B).astype(np.int64)
        if seed % 3 == 0:
            sizes[:] = sizes[0]          # ties: the stable order by index decides
        max_len = int(rng.choice([1000, 200, 64, 1]))
        for cls in (0, 1):
            lay = SeqLayout(_FakeGS(sizes), "packed", max_len, bool(cls))
            meta, desc, last, work = _host_layout(sizes, max_len, cls)
            assert meta == (lay.rows, lay.max_npos, lay.num_work)
            assert np.array_equal(desc, lay.desc_cpu)
            assert np.array_equal(last, lay.last_rows.numpy())
            assert np.array_equal(work, lay.work.numpy())


def test_host_layout_small_buffer_is_an_error():
    from graphtrans_amd import _lib
    sizes = np.array([5, 3, 9], np.int64)
    meta = (C.c_int64 * 6)()
    buf = np.zeros(8, np.uint8)
    rc = _lib.lib().gt_seq_layout_packed_host(sizes.ctypes.data, 3, 100, 1, buf.ctypes.data, buf.size, meta)
    assert rc != 0 and b"too small" in _lib.lib().gt_last_error()


def test_struct_mirrors_have_the_library_sizes():
    from graphtrans_amd import engine
    engine._ABI_OK.clear()
    engine._check_abi()   # raises on a mismatch
    assert engine._ABI_OK
