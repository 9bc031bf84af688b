"""GNNTransformer — GraphTrans (models/gnn_transformer.py:16-168) behind the reference's module
surface: same ctor `(num_tasks, node_encoder, edge_encoder_cls, args)`, `forward(batched_data,
perturb=None)`, static `add_args/name/get_emb_dim/need_deg`, `epoch_callback`, and the same
state_dict keys, so reference checkpoints load and `main.py` can construct it unchanged.

forward() picks between two equivalent data paths:
  * packed  (default when pooling is cls/last and neither pos_encoder nor the masked encoder is
             on): node rows -> token rows with NO padding rows at all; every Linear/LayerNorm/FFN
             and the attention kernel touch only real tokens.  Outputs equal the padded path's
             (padding never influences valid rows: masked keys, per-token FFN/LN).
  * padded  (the reference layout (S,B,d), exact for `mean` pooling which sums padded rows,
             gnn_transformer.py:117): pad_batch -> [pos_encoder] -> [masked encoder] -> encoder.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import engine, ops
from ..modules.gnn_module import GNNNodeEmbedding, batch_structure
from ..modules.masked_transformer_encoder import MaskedOnlyTransformerEncoder
from ..modules.transformer_encoder import TransformerNodeEncoder
from ..modules.utils import pad_batch
from .base_model import BaseModel, StackedHeads, stacked_heads  # noqa: F401


class GNNTransformer(BaseModel):
    @staticmethod
    def get_emb_dim(args):
        return args.gnn_emb_dim

    @staticmethod
    def add_args(parser):
        TransformerNodeEncoder.add_args(parser)
        MaskedOnlyTransformerEncoder.add_args(parser)
        group = parser.add_argument_group("GNNTransformer - Training Config")
        group.add_argument("--pos_encoder", default=False, action="store_true")
        group.add_argument("--pretrained_gnn", type=str, default=None, help="pretrained gnn_node node embedding path")
        group.add_argument("--freeze_gnn", type=int, default=None, help="Freeze gnn_node weight from epoch `freeze_gnn`")

    @staticmethod
    def name(args):
        name = f"{args.model_type}-pooling={args.graph_pooling}"
        name += "-norm_input" if args.transformer_norm_input else ""
        name += f"+{args.gnn_type}"
        name += "-virtual" if args.gnn_virtual_node else ""
        name += f"-JK={args.gnn_JK}"
        name += f"-enc_layer={args.num_encoder_layers}"
        name += f"-enc_layer_masked={args.num_encoder_layers_masked}"
        name += f"-d={args.d_model}"
        name += f"-act={args.transformer_activation}"
        name += f"-tdrop={args.transformer_dropout}"
        name += f"-gdrop={args.gnn_dropout}"
        name += "-pretrained_gnn" if args.pretrained_gnn else ""
        name += f"-freeze_gnn={args.freeze_gnn}" if args.freeze_gnn is not None else ""
        name += "-prenorm" if args.transformer_prenorm else "-postnorm"
        return name

    def __init__(self, num_tasks, node_encoder, edge_encoder_cls, args):
        super().__init__()
        self.gnn_node = GNNNodeEmbedding(args.gnn_virtual_node, args.gnn_num_layer, args.gnn_emb_dim, node_encoder,
                                         edge_encoder_cls, JK=args.gnn_JK, drop_ratio=args.gnn_dropout,
                                         residual=args.gnn_residual, gnn_type=args.gnn_type)
        if getattr(args, "pretrained_gnn", None):
            state_dict = torch.load(args.pretrained_gnn)
            self.gnn_node.load_state_dict(self._gnn_node_state(state_dict["model"]))
        self.freeze_gnn = getattr(args, "freeze_gnn", None)

        gnn_emb_dim = 2 * args.gnn_emb_dim if args.gnn_JK == "cat" else args.gnn_emb_dim
        self.gnn2transformer = nn.Linear(gnn_emb_dim, args.d_model)
        self.pos_encoder = PositionalEncoding(args.d_model, dropout=0) if getattr(args, "pos_encoder", False) else None
        self.transformer_encoder = TransformerNodeEncoder(args)
        self.masked_transformer_encoder = MaskedOnlyTransformerEncoder(args)
        self.num_encoder_layers = args.num_encoder_layers
        self.num_encoder_layers_masked = args.num_encoder_layers_masked

        self.num_tasks = num_tasks
        self.pooling = args.graph_pooling
        self.graph_pred_linear_list = torch.nn.ModuleList()
        self.max_seq_len = args.max_seq_len
        output_dim = args.d_model
        if args.max_seq_len is None:
            self.graph_pred_linear = torch.nn.Linear(output_dim, self.num_tasks)
        else:
            for _ in range(args.max_seq_len):
                self.graph_pred_linear_list.append(torch.nn.Linear(output_dim, self.num_tasks))
        # "auto" | "packed" | "padded"
        self.layout = getattr(args, "token_layout", "auto")

    # ---------------------------------------------------------------------------------------
    def _use_packed(self):
        if self.layout == "padded":
            return False
        ok = (self.pooling in ("cls", "last") and self.pos_encoder is None and self.num_encoder_layers_masked == 0
              and self.num_encoder_layers > 0)
        if self.layout == "packed" and not ok:
            raise ValueError("token_layout='packed' needs cls/last pooling, no pos_encoder, no masked encoder")
        return ok

    def forward(self, batched_data, perturb=None):
        if batched_data.batch.numel() == 0:
            raise ValueError("empty batch")
        if perturb is not None and perturb.shape[0] != batched_data.batch.numel():
            raise ValueError("perturb must have one row per node")
        enc = self.transformer_encoder
        max_len = int(enc.max_input_len)
        if engine.eligible(self, batched_data, perturb):  # whole model as one autograd node (engine.py)
            out = engine.forward(self, batched_data)   # graph structure, token layout, forward: one C call (csrc/model.hip)
            if self.max_seq_len is None:
                return out
            return StackedHeads(out.view(out.shape[0], self.max_seq_len, self.num_tasks))
        h_node = self.gnn_node(batched_data, perturb)
        h_node = ops.linear_module(self.gnn2transformer, h_node)
        gs = batch_structure(batched_data)

        if self._use_packed():
            with_cls = enc.cls_embedding is not None
            lay = gs.layout("packed", max_len, with_cls)
            tokens, _ = ops.seq_gather(h_node, enc.cls_embedding if with_cls else None, gs, lay)
            h_graph = enc.forward_tokens(tokens, lay, pooled=True).float()  # out[-1] of every sequence (the only rows read)
        else:
            padded_h_node, src_padding_mask, num_nodes, mask, max_num_nodes = pad_batch(
                h_node, batched_data.batch, max_len, get_mask=True, graph=gs)
            transformer_out = padded_h_node
            if self.pos_encoder is not None:
                transformer_out = self.pos_encoder(transformer_out)
            if self.num_encoder_layers_masked > 0:
                adj_list = batched_data.adj_list
                padded_adj_list = torch.zeros((len(adj_list), max_num_nodes, max_num_nodes), device=h_node.device)
                for idx, adj_list_item in enumerate(adj_list):  # top-left placement, as the reference (:104-107)
                    N, _ = adj_list_item.shape
                    padded_adj_list[idx, 0:N, 0:N] = torch.from_numpy(np.asarray(adj_list_item)).to(h_node.device)
                transformer_out = self.masked_transformer_encoder(
                    transformer_out.transpose(0, 1), attn_mask=padded_adj_list, valid_input_mask=src_padding_mask
                ).transpose(0, 1)
            if self.num_encoder_layers > 0:
                transformer_out, _ = enc(transformer_out, src_padding_mask)
            transformer_out = transformer_out.float()
            if self.pooling in ["last", "cls"]:
                h_graph = transformer_out[-1]
            elif self.pooling == "mean":  # divides by the number of PADDED positions (:117)
                h_graph = transformer_out.sum(0) / src_padding_mask.sum(-1, keepdim=True)
            else:
                raise NotImplementedError

        if self.max_seq_len is None:
            return ops.linear_module(self.graph_pred_linear, h_graph)
        return stacked_heads(h_graph, self.graph_pred_linear_list, self.num_tasks)

    def epoch_callback(self, epoch):
        if self.freeze_gnn is not None and epoch >= self.freeze_gnn:
            for param in self.gnn_node.parameters():
                param.requires_grad = False
            engine.invalidate(self)   # the fused path covers fully trainable models only

    def _gnn_node_state(self, state_dict):
        module_name = "gnn_node"
        new_state_dict = dict()
        for k, v in state_dict.items():
            if module_name in k:
                new_key = k.split(".")
                module_index = new_key.index(module_name)
                new_state_dict[".".join(new_key[module_index + 1:])] = v
        return new_state_dict


class PositionalEncoding(nn.Module):
    """models/gnn_transformer.py:149-168."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        pe[:, 0, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def forward(self, x):
        x = x + self.pe[: x.size(0)]
        return self.dropout(x)
