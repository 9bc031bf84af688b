"""CausalSelfAttention / Block / MaskedTransformerBlock / MaskedOnlyTransformerEncoder
(modules/masked_transformer_encoder.py:10-130): same parameters and state_dict keys
(`attn.{key,query,value,proj}`, `ln1`, `ln2`, `mlp.{0,2}`), attention core on the fused HIP kernel.

The reference materialises (B, nh, T, T) scores, fills them where the dense `attn_mask` (B,T,T) or
the per-key `valid_input_mask` (B,T) is 0 with the finite value -1e6 and soft-maxes; here q, k, v
are projected by the HIP linear kernel into one (B*T, 3C) buffer and gt_attn_fwd/bwd applies the
same masked_fill semantics in-register (a fully masked row becomes uniform, filled scores carry no
gradient).  No shipped reference config enables this encoder (`num_encoder_layers_masked`
defaults to 0, :108); `transformer_prenorm` is not forwarded by the reference (:114-121), so the
blocks are always pre-norm here too.
"""
import math

import torch
import torch.nn as nn

from .. import ops


class _BatchLayout:
    """B sequences of exactly T positions, batch-major rows (row = b*T + t): all keys in range."""

    def __init__(self, B, T, device):
        ar = torch.arange(B, dtype=torch.int32, device=device)
        self.desc = torch.stack([ar * T, torch.full_like(ar, T), torch.zeros_like(ar), torch.full_like(ar, T)], dim=1).contiguous()
        self.B, self.row_stride, self.rows, self.max_npos = B, 1, B * T, T


class CausalSelfAttention(nn.Module):
    def __init__(self, n_embd, n_head, attn_pdrop, resid_pdrop):
        super().__init__()
        assert n_embd % n_head == 0
        self.key = nn.Linear(n_embd, n_embd)
        self.query = nn.Linear(n_embd, n_embd)
        self.value = nn.Linear(n_embd, n_embd)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.resid_drop = nn.Dropout(resid_pdrop)
        self.proj = nn.Linear(n_embd, n_embd)
        self.n_head = n_head

    def forward(self, x, attn_mask: torch.Tensor = None, valid_input_mask: torch.Tensor = None, mask_value=-1e6):
        """x (B,T,C) batch-first; attn_mask (B,T,T): 0 = fill; valid_input_mask (B,T): 0 = fill that key."""
        B, T, C = x.size()
        rows = x.reshape(B * T, C)
        qkv = torch.cat([ops.linear_module(self.query, rows), ops.linear_module(self.key, rows),
                         ops.linear_module(self.value, rows)], dim=1)
        p = self.attn_drop.p if self.training else 0.0
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) if p > 0 else 0
        y = ops.attention(qkv, _BatchLayout(B, T, x.device), self.n_head, dropout_p=p, seed=seed,
                          scale=1.0 / math.sqrt(C // self.n_head), dense_mask=attn_mask, key_valid=valid_input_mask,
                          mask_value=mask_value)
        return ops.dropout(ops.linear_module(self.proj, y), self.resid_drop.p, self.training).view(B, T, C)   # resid_drop (:54)


class Block(nn.Module):
    def __init__(self, n_embd, n_ff, n_head, attn_pdrop, resid_pdrop, prenorm=True):
        super().__init__()
        self.prenorm = prenorm
        self.ln1 = nn.LayerNorm(n_embd)
        self.ln2 = nn.LayerNorm(n_embd)
        self.attn = CausalSelfAttention(n_embd, n_head, attn_pdrop, resid_pdrop)
        self.mlp = nn.Sequential(nn.Linear(n_embd, n_ff), nn.GELU(), nn.Linear(n_ff, n_embd), nn.Dropout(resid_pdrop))

    def _ln(self, ln, x, resid=None):
        shape = x.shape
        y = ops.layer_norm(x.reshape(-1, shape[-1]), ln.weight, ln.bias, ln.eps,
                           resid=None if resid is None else resid.reshape(-1, shape[-1]))
        return y.view(shape)

    def _mlp(self, x):
        shape = x.shape
        h = ops.linear_module(self.mlp[0], x.reshape(-1, shape[-1]), act="gelu")   # GELU in the GEMM epilogue
        return ops.dropout(ops.linear_module(self.mlp[2], h), self.mlp[3].p, self.training).view(shape)   # mlp[3] = nn.Dropout (:75)

    def forward(self, x, attn_mask=None, valid_input_mask=None):
        if self.prenorm:
            x = x + self.attn(self._ln(self.ln1, x), attn_mask, valid_input_mask)
            x = x + self._mlp(self._ln(self.ln2, x))
        else:  # post-norm: LN(x + sublayer(x)), the residual add fused into the LayerNorm kernel
            x = self._ln(self.ln1, self.attn(x, attn_mask, valid_input_mask), resid=x)
            x = self._ln(self.ln2, self._mlp(x), resid=x)
        return x


class MaskedTransformerBlock(nn.Module):
    def __init__(self, n_layer, n_embd, n_ff, n_head, attn_pdrop, resid_pdrop, prenorm=True):
        super().__init__()
        self.blocks = nn.ModuleList([Block(n_embd, n_ff, n_head, attn_pdrop, resid_pdrop, prenorm) for _ in range(n_layer)])

    def forward(self, x, attn_mask=None, valid_input_mask=None):
        for block in self.blocks:
            x = block(x, attn_mask, valid_input_mask)
        return x


class MaskedOnlyTransformerEncoder(nn.Module):
    @staticmethod
    def add_args(parser):
        group = parser.add_argument_group("Masked Transformer Encoder -- architecture config")
        group.add_argument("--num_encoder_layers_masked", type=int, default=0)
        group.add_argument("--transformer_prenorm", action="store_true", default=False)

    def __init__(self, args):
        super().__init__()
        self.max_input_len = args.max_input_len
        self.masked_transformer = MaskedTransformerBlock(args.num_encoder_layers_masked, args.d_model,
                                                         args.dim_feedforward, args.nhead, args.transformer_dropout,
                                                         args.transformer_dropout)

    def forward(self, x, attn_mask=None, valid_input_mask=None):
        """x (B,T,C); masks as CausalSelfAttention."""
        return self.masked_transformer(x.contiguous(), attn_mask=attn_mask, valid_input_mask=valid_input_mask)
