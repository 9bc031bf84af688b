"""bf16x3 weight images for the fp32-accurate GEMMs on the bf16 matrix pipe (csrc/linear3x.h, include/graphtrans_hip.h
"gt_w3_*").  The reference's big Linear layers (GCNConv.linear, the GIN / virtual-node MLPs, gnn2transformer:
modules/conv.py:44,51, modules/gnn_module.py:161-170, models/gnn_transformer.py:69-70) are fp32; an fp32 weight is split once
per optimizer step into three bf16 planes laid out in LDS order, and every big-M fp32 GEMM on a BOUND weight then runs six bf16
MFMA products per fp32-accurate product instead of the 16 x slower fp32 MFMA.

    imgs = W3Images([lin.weight, ...])     # one device buffer, one launch per build
    imgs.build()                           # after every optimizer step (weights changed)
    with imgs.bound(): y = ops.linear(x, lin.weight, lin.bias)     # per host thread; autograd's backward thread binds again
"""
import contextlib
import ctypes as C
import os

import torch

from . import _lib
from .graph import _stream

# GT_F32_GEMM=exact keeps the exact-fp32 MFMA kernels (v_mfma_f32_16x16x4_f32) everywhere: the parity yardstick
# (the GEMMs by not binding images, attention through the library's "attn_f32_exact" option -- set_exact() does both at run time)
ENABLED = os.environ.get("GT_F32_GEMM", "split") != "exact"


def set_exact(on):
    """exact-fp32 MFMA everywhere (True) or fp32-accurate bf16x6 products (False, the default); -> the previous setting.  Models built
    before the call keep the images they hold: engine.plan() reads ENABLED when a plan is made."""
    global ENABLED
    prev = not ENABLED
    ENABLED = not on
    _lib.option_set("attn_f32_exact", 1 if on else 0)
    return prev

# Images are a cache of the weights: torch's in-place ops bump a parameter's `_version`, kernels that write parameters through raw
# pointers (optim.FusedAdamW) do not -- they call weights_changed() instead; holders compare (versions, EPOCH).
EPOCH = 0


def weights_changed():
    global EPOCH
    EPOCH += 1


class W3Images:
    _fn = ("gt_w3_image_bytes", "gt_w3_images", "gt_w3_bind", "gt_w3_unbind")

    def __init__(self, weights, forward=True, transposed=True):
        self.weights = [w for w in weights]
        if not self.weights:
            raise ValueError("no weights")
        lib = _lib.lib()
        dev = self.weights[0].device
        jobs, off = [], 0
        self.fwd_off, self.t_off = [], []
        for w in self.weights:
            if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()):
                raise TypeError("W3Images: contiguous fp32 GPU matrices only")
            N, K = int(w.shape[0]), int(w.shape[1])
            fo = to = None
            nbytes = getattr(lib, self._fn[0])
            if forward and int(nbytes(N, K)):        # (0 bytes: a shape the kernel does not cover -- that direction stays unbound)
                fo, off = off, off + (int(nbytes(N, K)) + 1023) // 1024 * 1024
                jobs.append((w, N, K, 0, fo))
            if transposed and int(nbytes(K, N)):
                to, off = off, off + (int(nbytes(K, N)) + 1023) // 1024 * 1024
                jobs.append((w, N, K, 1, to))
            self.fwd_off.append(fo)
            self.t_off.append(to)
        self.buf = torch.empty(off + 1024, dtype=torch.uint8, device=dev)
        self.base = (self.buf.data_ptr() + 1023) // 1024 * 1024
        n = len(jobs)
        self._n = n
        self._w = (C.c_void_p * n)(*[j[0].data_ptr() for j in jobs])
        self._N = (C.c_int64 * n)(*[j[1] for j in jobs])
        self._K = (C.c_int64 * n)(*[j[2] for j in jobs])
        self._T = (C.c_int * n)(*[j[3] for j in jobs])
        self._img = (C.c_void_p * n)(*[self.base + j[4] for j in jobs])
        m = len(self.weights)
        self._bw = (C.c_void_p * m)(*[w.data_ptr() for w in self.weights])
        self._bN = (C.c_int64 * m)(*[int(w.shape[0]) for w in self.weights])
        self._bK = (C.c_int64 * m)(*[int(w.shape[1]) for w in self.weights])
        self._bf = (C.c_void_p * m)(*[(self.base + o) if o is not None else None for o in self.fwd_off])
        self._bt = (C.c_void_p * m)(*[(self.base + o) if o is not None else None for o in self.t_off])
        self.ptrs = tuple(w.data_ptr() for w in self.weights)

    def current(self):
        """False once a weight's storage moved (then build a new W3Images)"""
        return self.ptrs == tuple(w.data_ptr() for w in self.weights)

    def build(self, stream=None):
        if self._n:
            _lib.check(getattr(_lib.lib(), self._fn[1])(self._n, self._w, self._N, self._K, self._T, self._img,
                                                        _stream() if stream is None else stream), self._fn[1])

    def bind(self):
        _lib.check(getattr(_lib.lib(), self._fn[2])(len(self.weights), self._bw, self._bN, self._bK, self._bf, self._bt), self._fn[2])

    @classmethod
    def unbind(cls):
        getattr(_lib.lib(), cls._fn[3])()

    @contextlib.contextmanager
    def bound(self):
        self.bind()
        try:
            yield self
        finally:
            self.unbind()


# GT_BF16_GEMM=tiled keeps the tiled bf16 kernels for the encoder layers (A/B yardstick)
W1_ENABLED = os.environ.get("GT_BF16_GEMM", "stationary") != "tiled"


class W1Images(W3Images):
    """bf16 images in MFMA fragment order for the encoder layers' GEMMs (csrc/linear1.h, "gt_w1_*"): nn.TransformerEncoderLayer's
    in_proj / out_proj / linear1 / linear2 (modules/transformer_encoder.py:28-32) with bf16 token rows run with the weight stationary
    in registers on a BOUND weight; same life cycle as W3Images (build after every optimizer step, bind per host thread)."""
    _fn = ("gt_w1_image_bytes", "gt_w1_images", "gt_w1_bind", "gt_w1_unbind")
