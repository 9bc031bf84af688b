"""TransformerNodeEncoder (modules/transformer_encoder.py:9-61): same flags, parameters and
state_dict keys (the torch nn.TransformerEncoder modules are kept as parameter containers), but
the forward runs our own post-norm encoder on token rows with the fused HIP attention kernel.

Two entry points:
  forward(padded_h_node, src_padding_mask)  -- the reference signature, (S,B,d) + (B,S) mask
  forward_tokens(tokens, layout)            -- token rows in any SeqLayout (packed = no padding)
"""
import numpy as np
import torch
import torch.nn as nn

from .. import layers, ops


class _MaskLayout:
    """SeqLayout built on device from a (B,S) key-padding mask whose valid keys are contiguous
    (what pad_batch produces: left padding).  No host sync."""

    def __init__(self, mask, S, B):
        valid = ~mask
        kv_len = valid.sum(1).to(torch.int32)
        kv_off = valid.to(torch.int32).argmax(1).to(torch.int32)
        desc = torch.stack([torch.arange(B, dtype=torch.int32, device=mask.device),
                            torch.full((B,), S, dtype=torch.int32, device=mask.device), kv_off, kv_len], dim=1)
        self.desc = desc.contiguous()
        self.kind, self.row_stride, self.rows, self.max_npos, self.B, self.S = "padded", B, S * B, S, B, S


class TransformerNodeEncoder(nn.Module):
    @staticmethod
    def add_args(parser):
        group = parser.add_argument_group("transformer")
        group.add_argument("--d_model", type=int, default=128, help="transformer d_model.")
        group.add_argument("--nhead", type=int, default=4, help="transformer heads")
        group.add_argument("--dim_feedforward", type=int, default=512, help="transformer feedforward dim")
        group.add_argument("--transformer_dropout", type=float, default=0.3)
        group.add_argument("--transformer_activation", type=str, default="relu")
        group.add_argument("--num_encoder_layers", type=int, default=4)
        group.add_argument("--max_input_len", default=1000, help="The max input length of transformer input")
        group.add_argument("--transformer_norm_input", action="store_true", default=False)

    def __init__(self, args):
        super().__init__()
        self.d_model = args.d_model
        self.num_layer = args.num_encoder_layers
        self.nhead = args.nhead
        self.dropout_p = float(args.transformer_dropout)
        self.activation = args.transformer_activation
        encoder_layer = nn.TransformerEncoderLayer(args.d_model, args.nhead, args.dim_feedforward,
                                                   args.transformer_dropout, args.transformer_activation)
        encoder_norm = nn.LayerNorm(args.d_model)
        # parameter container only (same keys / deep-copied init as the reference); never called
        self.transformer = nn.TransformerEncoder(encoder_layer, args.num_encoder_layers, encoder_norm,
                                                 enable_nested_tensor=False)
        self.max_input_len = args.max_input_len
        self.norm_input = None
        if args.transformer_norm_input:
            self.norm_input = nn.LayerNorm(args.d_model)
        self.cls_embedding = None
        if args.graph_pooling == "cls":
            self.cls_embedding = nn.Parameter(torch.randn([1, 1, args.d_model], requires_grad=True))
        # storage/compute dtype of the token stream: torch.float32 (parity) or torch.bfloat16 (MFMA bf16)
        self.compute_dtype = getattr(args, "compute_dtype", torch.float32)

    # ---- building blocks -------------------------------------------------------------------
    def _w(self, p):
        return p if p.dtype == self.compute_dtype else p.to(self.compute_dtype)

    def _ln(self, x, ln, resid=None, seed=0):
        """LN(resid + dropout(x)) in one kernel (gt_layernorm_fwd); plain LN when resid is None."""
        p = self.dropout_p if (self.training and resid is not None) else 0.0
        return ops.layer_norm(x, ln.weight, ln.bias, ln.eps, resid=resid, dropout_p=p, seed=seed)

    def _layer(self, x, mod, lay, seed):
        sa = mod.self_attn
        p = self.dropout_p if self.training else 0.0
        if layers.encoder_layer_eligible(mod, x, self.activation):  # one composite op per layer
            return layers.encoder_layer(x, mod, lay, self.nhead, p, seed, self.training, self.activation)
        qkv = ops.linear(x, sa.in_proj_weight, sa.in_proj_bias)
        ctx = ops.attention(qkv, lay, self.nhead, dropout_p=p, seed=seed)
        a = ops.linear(ctx, sa.out_proj.weight, sa.out_proj.bias)
        x = self._ln(a, mod.norm1, resid=x, seed=seed ^ 0x5851F42D4C957F2D)
        # activation (relu / gelu) + dropout fused into linear1's epilogue
        f = ops.linear(x, mod.linear1.weight, mod.linear1.bias, act=self.activation, dropout_p=p, seed=seed ^ 0x2545F4914F6CDD1D)
        f = ops.linear(f, mod.linear2.weight, mod.linear2.bias)
        return self._ln(f, mod.norm2, resid=x, seed=seed ^ 0x14057B7EF767814F)

    def forward_tokens(self, tokens, lay, pooled=False):
        """tokens (lay.rows, d) already containing the CLS rows -> (lay.rows, d).
        pooled=True (cls / last pooling: only transformer_out[-1] of every sequence is read, models/gnn_transformer.py:113-114) ->
        (lay.B, d), the output in the LAST position of every sequence: the last layer then computes keys and values for every row
        but attention, out_proj, the FFN and both norms for the pooled rows only (layers.encoder_layer_pooled), and the final norm
        runs on those B rows.  Same values as the full computation followed by the row selection (at dropout 0)."""
        x = tokens.to(self.compute_dtype)
        if self.norm_input is not None:
            x = self._ln(x, self.norm_input)
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) if (self.training and self.dropout_p > 0) else 0
        nl = len(self.transformer.layers)
        done = False
        for i, mod in enumerate(self.transformer.layers):
            s_i = (seed + 0x9E3779B97F4A7C15 * (i + 1)) & 0xFFFFFFFFFFFFFFFF
            if pooled and i == nl - 1 and layers.encoder_layer_eligible(mod, x, self.activation) and lay.kind == "packed":
                p = self.dropout_p if self.training else 0.0
                x = layers.encoder_layer_pooled(x, mod, lay, self.nhead, p, s_i, self.training, self.activation)
                done = True
            else:
                x = self._layer(x, mod, lay, s_i)
        if pooled and not done:
            x = x.index_select(0, lay.last_rows)
        if pooled:
            x = x.float()   # the final norm on fp32 copies of the pooled rows (what the fused path does)
        if self.transformer.norm is not None:
            x = self._ln(x, self.transformer.norm)
        return x

    def forward(self, padded_h_node, src_padding_mask):
        """padded_h_node: (S,B,d); src_padding_mask: (B,S) True = padding  ->  ((S',B,d), (B,S'))"""
        if self.cls_embedding is not None:
            expand_cls = self.cls_embedding.expand(1, padded_h_node.size(1), -1).to(padded_h_node.dtype)
            padded_h_node = torch.cat([padded_h_node, expand_cls], dim=0)
            zeros = src_padding_mask.data.new(src_padding_mask.size(0), 1).fill_(0)
            src_padding_mask = torch.cat([src_padding_mask, zeros], dim=1)
        S, B, d = padded_h_node.shape
        lay = _MaskLayout(src_padding_mask, S, B)
        out = self.forward_tokens(padded_h_node.reshape(S * B, d), lay)
        return out.view(S, B, d), src_padding_mask
