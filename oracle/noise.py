"""oracle/noise.py — the oracle's OWN sensitivity to reduced-precision GEMM operands, per gradient tensor.

*** TEST INFRASTRUCTURE (tests/, bench.py's precision_vs_oracle leg); never on the product path. ***

The reduced-precision modes of the HIP path ("mixed": bf16 token rows + bf16 MFMA inside the encoder layers, "bf16": bf16
MFMA operands in every GEMM) cannot be held to the fp32 bar; a blanket bound per mode (round 2: worst tensor <= 0.2 / 0.8)
is wide enough to hide a wrong tensor.  What a bf16 operand does to the math is known, though: it moves each GEMM operand
element by a relative error of up to 2^-9.  `lowp_noise` measures what THAT does to every parameter gradient of the
float64 oracle (reference math, no kernel involved): the oracle is re-run with the GEMM weights of the affected layers
multiplied by (1 + 2^-9 u), u ~ U(-1, 1), and the relative L2 distance of each gradient tensor to the unperturbed float64
gradient is the tensor's noise floor.  An ill-conditioned gradient (GINConv's `eps`, modules/conv.py:21,28 of the
reference: one scalar = a sum of N x D products of both signs) shows up with a large floor because the ORACLE moves that
much -- and a HIP gradient is then bounded by a small multiple of its own tensor's floor instead of a per-mode constant.
Round 4: activations too.  The reduced-precision modes also ROUND activations -- the encoder's token-row tensors are stored in
bf16 (mixed, bf16) and every GEMM's row operand is rounded on the fly (bf16) -- and their gradients likewise.  reference_math
carries identity taps at those points (`_tap(x, kind)`); `lowp_noise(..., activations=True)` installs a tap that multiplies the
forward value AND the gradient flowing back through it by (1 + 2^-9 u), fresh u per tensor.  With both operand classes perturbed
the floor is what the mode's arithmetic does to the oracle, and the bound on a HIP gradient is a small multiple of it without an
absolute floor doing the work (12.8 x the weights-only floor on Code2 mixed).
"""
import copy
from types import SimpleNamespace

import torch

BF16_EPS = 2.0 ** -9   # half an ulp of an 8-bit significand, relative


def gemm_weight_keys(sd, mode):
    """state_dict keys of the GEMM weights that the HIP path rounds to bf16 in `mode`"""
    keys = []
    for k, v in sd.items():
        if not (torch.is_tensor(v) and v.is_floating_point() and v.dim() == 2 and k.endswith("weight")):
            continue
        if any(t in k for t in ("node_encoder", "edge_encoder", "virtualnode_embedding", "bond_embedding", "atom_embedding")):
            continue   # embedding tables are gathered, not multiplied; the K <= 4 edge Linear runs in fp32 registers in every mode
        in_encoder = k.startswith("transformer_encoder.transformer.layers.")
        if mode == "mixed" and not in_encoder:
            continue
        keys.append(k)
    return keys


class _RoundLike(torch.autograd.Function):
    """x -> x (1 + rel u) forward, g -> g (1 + rel u') backward: a stored activation and its stored gradient, both rounded"""

    @staticmethod
    def forward(ctx, x, gen, rel):
        ctx.gen, ctx.rel = gen, rel
        return x * (1.0 + rel * (2.0 * torch.rand(x.shape, generator=gen, dtype=x.dtype) - 1.0))

    @staticmethod
    def backward(ctx, g):
        return g * (1.0 + ctx.rel * (2.0 * torch.rand(g.shape, generator=ctx.gen, dtype=g.dtype) - 1.0)), None, None


def lowp_noise(sd64, oargs, batch, fwd, loss_of, ref_g64, mode, seeds=(11, 12, 13, 14), rel=BF16_EPS, activations=True):
    """-> {gradient key: max over seeds of ||g_perturbed - g64|| / ||g64||}.  sd64: float64 state dict (leaf tensors),
    fwd: oracle.reference_math.gnn_transformer / pna_transformer, loss_of(outputs) -> scalar.  activations: also perturb the
    activations the mode rounds (and their gradients) through reference_math's storage taps."""
    from . import reference_math as rm
    keys = gemm_weight_keys(sd64, mode)
    kinds = ("enc",) if mode == "mixed" else ("enc", "gemm_in")
    noise = {k: 0.0 for k in ref_g64}
    b64 = copy.copy(batch)   # dense float inputs (TU / ER node features, Linear edge attributes) in the oracle's float64
    for name in ("x", "edge_attr"):
        v = getattr(b64, name, None)
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(b64, name, v.double())
    batch = b64
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        for seed in seeds:
            g = torch.Generator().manual_seed(seed)
            sd = {}
            for k, v in sd64.items():
                if torch.is_tensor(v) and v.is_floating_point():
                    w = v.detach().clone()
                    if k in keys:
                        w = w * (1.0 + rel * (2.0 * torch.rand(w.shape, generator=g, dtype=torch.float64) - 1.0))
                    sd[k] = w.requires_grad_(True)
                else:
                    sd[k] = v.clone() if torch.is_tensor(v) else copy.copy(v)
            if activations:
                rm._TAP = lambda x, kind, g_=g: _RoundLike.apply(x, g_, rel) if (kind in kinds and x.requires_grad) else x
            try:
                loss_of(fwd(sd, oargs, batch, None, True)).backward()
            finally:
                rm._TAP = None
            for k, r in ref_g64.items():
                gk = sd[k].grad
                if gk is None:
                    continue
                n = float(r.norm())
                if n > 0:
                    noise[k] = max(noise[k], float((gk - r).norm()) / n)
    finally:
        torch.set_default_dtype(old)
    return noise


def oracle_args(args):
    return SimpleNamespace(**{k: v for k, v in vars(args).items() if k not in ("compute_dtype", "token_layout")})
