"""GNN_node / GNN_node_Virtualnode / GNNNodeEmbedding (modules/gnn_module.py:18-248) on the HIP
aggregate and segment kernels.  Same ctor arguments, attributes and state_dict keys."""
import torch
import torch.nn.functional as F

from .. import layers, ops
from ..graph import GraphStructure
from .conv import GCNConv, GINConv, edge_spec
from .norm import BatchNorm1d, mlp_bn_relu


def batch_structure(batched_data):
    """GraphStructure of a collated batch, built once and cached on the batch object (the fused model path builds its own inside
    the driver's arena, csrc/model.hip, and reuses a cached one when it finds it)."""
    gs = getattr(batched_data, "_gt_structure", None)
    if gs is None:
        sizes = getattr(batched_data, "_sizes", None)
        ng = getattr(batched_data, "_num_graphs", None)
        if ng is None and sizes is None and hasattr(type(batched_data), "num_graphs"):
            try:
                ng = batched_data.num_graphs
            except Exception:
                ng = None
        gs = GraphStructure.build(batched_data.edge_index, batched_data.batch, num_graphs=ng, sizes=sizes)
        try:
            batched_data._gt_structure = gs
        except Exception:
            pass
    return gs


def _encode_nodes(node_encoder, batched_data):
    x = batched_data.x
    node_depth = batched_data.node_depth if hasattr(batched_data, "node_depth") else None
    if node_encoder is None:
        return x
    if type(node_encoder) is torch.nn.Linear and node_depth is None and x.is_cuda and x.dim() == 2:
        return ops.linear_module(node_encoder, x)   # TU datasets: nn.Linear(F, D) (dataset/tud.py:65) on the HIP GEMM
    return node_encoder(x) if node_depth is None else node_encoder(x, node_depth.view(-1))


def _make_convs(self, num_layer, emb_dim, edge_encoder_cls, gnn_type):
    self.convs = torch.nn.ModuleList()
    self.batch_norms = torch.nn.ModuleList()
    for _ in range(num_layer):
        if gnn_type == "gin":
            self.convs.append(GINConv(emb_dim, edge_encoder_cls))
        elif gnn_type == "gcn":
            self.convs.append(GCNConv(emb_dim, edge_encoder_cls))
        else:
            raise ValueError("Undefined GNN type called {}".format(gnn_type))
        self.batch_norms.append(BatchNorm1d(emb_dim))


_SEED_STEP = 0x9E3779B97F4A7C15
_SEED_MASK = 0xFFFFFFFFFFFFFFFF


def _gnn_seed(self):
    """Base seed of this forward's GNN dropout masks (one draw from torch's CPU generator)."""
    if self.training and self.drop_ratio > 0:
        return int(torch.empty((), dtype=torch.int64).random_().item())
    return 0


def layer_seed(base, layer):
    return (base + _SEED_STEP * (2 * layer + 1)) & _SEED_MASK


def vn_seed(base, layer):
    return (base + _SEED_STEP * (2 * layer + 2)) & _SEED_MASK


def _conv_bn_layer(self, layer, h_list, vn, gs, edge_index, edge_attr, base_seed=0):
    """One message-passing layer: [x = h + vn[batch]] -> conv -> BN (+ReLU except on the last layer)
    -> dropout -> [+ x].  Returns (x, h).  GCN layers in the covered configuration run as ONE
    composite op (layers.gcn_layer); everything else through the fine-grained ops."""
    conv, bn = self.convs[layer], self.batch_norms[layer]
    relu = layer != self.num_layer - 1
    h_in = h_list[layer]
    p = self.drop_ratio if self.training else 0.0
    seed = layer_seed(base_seed, layer)
    if isinstance(conv, GCNConv):
        spec = edge_spec(conv.edge_encoder, edge_attr, conv.emb_dim)
        if layers.gcn_layer_eligible(conv, bn, h_in, spec, self.drop_ratio, self.training):
            return layers.gcn_layer(h_in, vn, gs, conv, bn, spec, relu, self.residual, self.training, p, seed)
    x = ops.segment_bcast_add(h_in, vn, gs) if vn is not None else h_in  # + vn[batch]   (:199)
    h = conv(x, edge_index, edge_attr, graph=gs)
    h = bn(h, relu=relu, dropout_p=p, seed=seed)  # F.dropout(h, drop_ratio) fused behind the BatchNorm (:88-90,209-212)
    if self.residual:
        h = h + x
    return x, h


def _jk(JK, h_list, num_layer):
    if JK == "last":
        return h_list[-1]
    if JK == "sum":  # excludes the final layer's output, as the reference does (gnn_module.py:100-103)
        out = 0
        for layer in range(num_layer):
            out = out + h_list[layer]
        return out
    if JK == "cat":
        return torch.cat([h_list[0], h_list[-1]], dim=-1)
    raise ValueError(JK)


class GNN_node(torch.nn.Module):
    @staticmethod
    def need_deg():
        return False

    def __init__(self, num_layer, emb_dim, node_encoder, edge_encoder_cls, drop_ratio=0.5, JK="last", residual=False,
                 gnn_type="gin"):
        super().__init__()
        self.num_layer, self.drop_ratio, self.JK, self.residual = num_layer, drop_ratio, JK, residual
        if self.num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")
        self.node_encoder = node_encoder
        _make_convs(self, num_layer, emb_dim, edge_encoder_cls, gnn_type)

    def forward(self, batched_data, perturb=None):
        gs = batch_structure(batched_data)
        edge_index, edge_attr = batched_data.edge_index, batched_data.edge_attr
        encoded = _encode_nodes(self.node_encoder, batched_data)
        h_list = [encoded + perturb if perturb is not None else encoded]
        base = _gnn_seed(self)
        for layer in range(self.num_layer):
            h_list.append(_conv_bn_layer(self, layer, h_list, None, gs, edge_index, edge_attr, base)[1])
        return _jk(self.JK, h_list, self.num_layer)


class GNN_node_Virtualnode(torch.nn.Module):
    @staticmethod
    def need_deg():
        return False

    def __init__(self, num_layer, emb_dim, node_encoder, edge_encoder_cls, drop_ratio=0.5, JK="last", residual=False,
                 gnn_type="gin"):
        super().__init__()
        self.num_layer, self.drop_ratio, self.JK, self.residual = num_layer, drop_ratio, JK, residual
        if self.num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")
        self.node_encoder = node_encoder
        self.virtualnode_embedding = torch.nn.Embedding(1, emb_dim)
        torch.nn.init.constant_(self.virtualnode_embedding.weight.data, 0)
        _make_convs(self, num_layer, emb_dim, edge_encoder_cls, gnn_type)
        self.mlp_virtualnode_list = torch.nn.ModuleList()
        for _ in range(num_layer - 1):
            self.mlp_virtualnode_list.append(torch.nn.Sequential(
                torch.nn.Linear(emb_dim, 2 * emb_dim), BatchNorm1d(2 * emb_dim), torch.nn.ReLU(),
                torch.nn.Linear(2 * emb_dim, emb_dim), BatchNorm1d(emb_dim), torch.nn.ReLU()))

    def forward(self, batched_data, perturb=None):
        gs = batch_structure(batched_data)
        edge_index, edge_attr = batched_data.edge_index, batched_data.edge_attr
        encoded = _encode_nodes(self.node_encoder, batched_data)
        h_list = [encoded + perturb if perturb is not None else encoded]
        # one zero-initialised embedding row per graph (gnn_module.py:195), without the .item() sync
        vn = self.virtualnode_embedding.weight.expand(gs.B, -1)
        base = _gnn_seed(self)
        p = self.drop_ratio if self.training else 0.0
        for layer in range(self.num_layer):
            h_list[layer], h = _conv_bn_layer(self, layer, h_list, vn, gs, edge_index, edge_attr, base)
            h_list.append(h)
            if layer < self.num_layer - 1:
                seq = self.mlp_virtualnode_list[layer]
                if layers.vn_update_eligible(seq, h_list[layer], self.drop_ratio, self.training):
                    vn = layers.vn_update(h_list[layer], vn, gs, seq, self.residual, self.training, p, vn_seed(base, layer))
                else:
                    t = ops.segment_sum(h_list[layer], gs, add=vn)  # global_add_pool + vn   (:219)
                    t = mlp_bn_relu(seq, t, dropout_p=p, seed=vn_seed(base, layer))  # F.dropout fused (:222)
                    vn = vn + t if self.residual else t
        return _jk(self.JK, h_list, self.num_layer)


def GNNNodeEmbedding(virtual_node, *args, **kwargs):
    if virtual_node:
        return GNN_node_Virtualnode(*args, **kwargs)
    return GNN_node(*args, **kwargs)
