"""CausalSelfAttention / Block / MaskedTransformerBlock / MaskedOnlyTransformerEncoder
(modules/masked_transformer_encoder.py:10-130), same parameters and state_dict keys.

No shipped reference config enables this branch (`num_encoder_layers_masked` defaults to 0,
masked_transformer_encoder.py:108).  Its attention takes a DENSE (B,T,T) mask with a finite fill
value (-1e6), which the range-masked HIP kernel does not cover yet: the score/softmax/PV core here
is composed from torch GPU ops (rocBLAS bmm) — documented in DESIGN.md as the one attention
variant not yet on a hand-written kernel.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


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
        B, T, C = x.size()
        hs = C // self.n_head
        k = self.key(x).view(B, T, self.n_head, hs).transpose(1, 2)
        q = self.query(x).view(B, T, self.n_head, hs).transpose(1, 2)
        v = self.value(x).view(B, T, self.n_head, hs).transpose(1, 2)
        att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hs))
        if attn_mask is not None:
            att = att.masked_fill(attn_mask.unsqueeze(1) == 0, mask_value)
        if valid_input_mask is not None:
            att = att.masked_fill(valid_input_mask.unsqueeze(1).unsqueeze(2) == 0, mask_value)
        att = self.attn_drop(F.softmax(att, dim=-1))
        y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)
        return self.resid_drop(self.proj(y))


class Block(nn.Module):
    def __init__(self, n_embd, n_ff, n_head, attn_pdrop, resid_pdrop, prenorm=True):
        super().__init__()
        self.prenorm = prenorm
        self.ln1 = nn.LayerNorm(n_embd)
        self.ln2 = nn.LayerNorm(n_embd)
        self.attn = CausalSelfAttention(n_embd, n_head, attn_pdrop, resid_pdrop)
        self.mlp = nn.Sequential(nn.Linear(n_embd, n_ff), nn.GELU(), nn.Linear(n_ff, n_embd), nn.Dropout(resid_pdrop))

    def forward(self, x, attn_mask=None, valid_input_mask=None):
        if self.prenorm:
            x = x + self.attn(self.ln1(x), attn_mask, valid_input_mask)
            x = x + self.mlp(self.ln2(x))
        else:
            x = self.ln1(x + self.attn(x, attn_mask, valid_input_mask))
            x = self.ln2(x + self.mlp(x))
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
        # `transformer_prenorm` is not forwarded by the reference either (:114-121): always pre-norm
        self.masked_transformer = MaskedTransformerBlock(args.num_encoder_layers_masked, args.d_model,
                                                         args.dim_feedforward, args.nhead, args.transformer_dropout,
                                                         args.transformer_dropout)

    def forward(self, x, attn_mask=None, valid_input_mask=None):
        return self.masked_transformer(x, attn_mask=attn_mask, valid_input_mask=valid_input_mask)
