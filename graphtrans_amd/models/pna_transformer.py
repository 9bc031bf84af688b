"""PNATransformer (models/pna_transformer.py:16-118): PNA node embedding -> Transformer, same
constructor / forward / flags / state_dict keys as the reference."""
import torch
import torch.nn as nn

from .. import ops
from ..modules.gnn_module import batch_structure
from ..modules.pna.pna_module import PNANodeEmbedding
from ..modules.transformer_encoder import TransformerNodeEncoder
from ..modules.utils import pad_batch
from .base_model import BaseModel, stacked_heads


class PNATransformer(BaseModel):
    @staticmethod
    def get_emb_dim(args):
        return args.gnn_emb_dim

    @staticmethod
    def need_deg():
        return True

    @staticmethod
    def add_args(parser):
        TransformerNodeEncoder.add_args(parser)
        PNANodeEmbedding.add_args(parser)
        group = parser.add_argument_group("GNNTransformer - Training Config")
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
        name += f"-d={args.d_model}"
        name += f"-act={args.transformer_activation}"
        name += f"-tdrop={args.transformer_dropout}"
        name += f"-gdrop={args.gnn_dropout}"
        name += "-pretrained_gnn" if args.pretrained_gnn else ""
        name += f"-freeze_gnn={args.freeze_gnn}" if args.freeze_gnn is not None else ""
        return name

    def __init__(self, num_tasks, node_encoder, edge_encoder_cls, args):
        super().__init__()
        self.gnn_node = PNANodeEmbedding(node_encoder, args)
        if getattr(args, "pretrained_gnn", None):
            state_dict = torch.load(args.pretrained_gnn)["model"]
            self.gnn_node.load_state_dict({k.split("gnn_node.", 1)[1]: v for k, v in state_dict.items() if "gnn_node." in k})
        self.freeze_gnn = getattr(args, "freeze_gnn", None)
        gnn_emb_dim = 2 * args.gnn_emb_dim if args.gnn_JK == "cat" else args.gnn_emb_dim
        self.gnn2transformer = nn.Linear(gnn_emb_dim, args.d_model)
        self.transformer_encoder = TransformerNodeEncoder(args)
        self.num_tasks = num_tasks
        self.pooling = args.graph_pooling
        self.graph_pred_linear_list = torch.nn.ModuleList()
        self.max_seq_len = args.max_seq_len
        if args.max_seq_len is None:
            self.graph_pred_linear = torch.nn.Linear(args.d_model, self.num_tasks)
        else:
            for _ in range(args.max_seq_len):
                self.graph_pred_linear_list.append(torch.nn.Linear(args.d_model, self.num_tasks))
        self.layout = getattr(args, "token_layout", "auto")

    def forward(self, batched_data, perturb=None):
        from .. import engine
        if engine.eligible(self, batched_data, perturb):   # whole model as one autograd node, one C call per direction (engine.py)
            out = engine.forward(self, batched_data)
            if self.max_seq_len is None:
                return out
            from .base_model import StackedHeads
            return StackedHeads(out.view(out.shape[0], self.max_seq_len, self.num_tasks))
        h_node = self.gnn_node(batched_data, perturb)
        h_node = ops.linear_module(self.gnn2transformer, h_node)
        gs = batch_structure(batched_data)
        enc = self.transformer_encoder
        max_len = int(enc.max_input_len)
        if self.pooling in ("cls", "last") and self.layout != "padded":
            with_cls = enc.cls_embedding is not None
            lay = gs.layout("packed", max_len, with_cls)
            tokens, _ = ops.seq_gather(h_node, enc.cls_embedding if with_cls else None, gs, lay)
            h_graph = enc.forward_tokens(tokens, lay, pooled=True).float()
        else:
            padded_h_node, src_padding_mask = pad_batch(h_node, batched_data.batch, max_len, graph=gs)
            transformer_out, mask = enc(padded_h_node, src_padding_mask)
            transformer_out = transformer_out.float()
            if self.pooling in ["last", "cls"]:
                h_graph = transformer_out[-1]
            elif self.pooling == "mean":  # valid-position count here (pna_transformer.py:89)
                h_graph = transformer_out.sum(0) / (~mask).sum(-1, keepdim=True)
            else:
                raise NotImplementedError
        if self.max_seq_len is None:
            return ops.linear_module(self.graph_pred_linear, h_graph)
        return stacked_heads(h_graph, self.graph_pred_linear_list, self.num_tasks)

    def epoch_callback(self, epoch):
        if self.freeze_gnn is not None and epoch >= self.freeze_gnn:
            for param in self.gnn_node.parameters():
                param.requires_grad = False
            from .. import engine
            engine.invalidate(self)   # the fused path covers fully trainable models only
